// microbench: v_fma_f32 vs v_pk_fma_f32 issue rate, and DPP fmac
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k_fma(int iters, float* out) {
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        }
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) out[0] = a0;
}
__global__ void k_pk(int iters, float* out) {
    f2 a0 = {(float)threadIdx.x, 1}, a1 = {1, 2}, a2 = {2, 3}, a3 = {3, 4}, a4 = {4, 5}, a5 = {5, 6}, a6 = {6, 7}, a7 = {7, 8}, b = {1.0001f, 1.0002f}, c = {0.5f, 0.25f};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        }
    }
    if (a0.x + a1.y + a2.x + a3.x + a4.x + a5.x + a6.x + a7.x == 12345.f) out[0] = a0.x;
}
int main() {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, blocks = 256 * 8, thr = 256;
    for (int rep = 0; rep < 2; rep++) {
        float ms;
        hipEventRecord(e0); hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(thr), 0, 0, iters, d); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        double n = (double)blocks * thr / 64 * iters * 64;   // wave instructions
        printf("v_fma_f32   : %.3f ms, %.2f Ginstr(wave)/s, %.1f TFLOP/s\n", ms, n / ms / 1e6, n * 64 * 2 / ms / 1e9);
        hipEventRecord(e0); hipLaunchKernelGGL(k_pk, dim3(blocks), dim3(thr), 0, 0, iters, d); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("v_pk_fma_f32: %.3f ms, %.2f Ginstr(wave)/s, %.1f TFLOP/s\n", ms, n / ms / 1e6, n * 64 * 4 / ms / 1e9);
    }
    return 0;
}
