// What does the fp16 matrix pipe sustain on operands that look like the split convolution's?  (round 5)
// Every wave loops over 12 v_mfma_f32_32x32x16_f16 on two accumulators, rotating through NSET operand register sets (so that consecutive
// MFMAs see different operand bits, as the convolution's K loop does), 4 waves per SIMD, for ~0.1 s per configuration so that the power
// management settles.  Operand kinds: constant (the guide's microbenchmark), random fp16 "high parts" (|x| <= 2^15, random mantissas and
// signs) and the h / l mix of the two-part form (the three terms a_h b_l, a_l b_h, a_h b_h in the kernel's order).
// Output: ns per MFMA per SIMD; 32 cycles at 2.4 GHz = 13.3 ns = the 2.5 PFLOP/s dense peak.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_floor_f16 mfma_floor_f16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int NSET = 4;

// kind 0: constants; 1: random high parts; 2: h / l mix
__global__ void __launch_bounds__(256) floor_kernel(int iters, int kind, float* out)
{
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
    f16x8 ah[NSET], al[NSET], bh[NSET], bl[NSET];
    uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int s = 0; s < NSET; s++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float xa = 1.0f + (threadIdx.x & 7) * 0.125f, xb = 0.5f;
            if (kind) {
                h = h * 1664525u + 1013904223u;
                xa = ((int)(h >> 8 & 0xffff) - 32768) * 0.5f;          // |x| <= 2^14: the scaled operand's range
                h = h * 1664525u + 1013904223u;
                xb = ((int)(h >> 8 & 0xffff) - 32768) * 0.5f + ((h >> 3) & 1023) * (1.0f / 1024.f);
                xa += ((h >> 13) & 1023) * (1.0f / 1024.f);
            }
            const _Float16 a_h = (_Float16)xa, b_h = (_Float16)xb;
            ah[s][i] = a_h; bh[s][i] = b_h;
            al[s][i] = kind == 2 ? (_Float16)(xa - (float)a_h) : a_h;
            bl[s][i] = kind == 2 ? (_Float16)(xb - (float)b_h) : b_h;
        }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int s = 0; s < NSET; s++) {
            // the kernel's order per K tile and pair of blocks: smallest terms first
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[(s + 1) % NSET], bl[s], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[(s + 1) % NSET], bh[s], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[(s + 1) % NSET], bh[s], acc1, 0, 0, 0);
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) r += acc0[i] + acc1[i];
    if (r == 12345.678f) out[0] = r;
}

int main()
{
    float* out;
    (void)hipMalloc(&out, 64);
    const char* names[3] = { "constant operands", "random high parts", "h / l mix (two-part form)" };
    for (int kind = 0; kind < 3; kind++)
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int blocks = 256 * wps;
            const int iters = 40000 / wps;                  // ~0.1-0.2 s per configuration
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            hipLaunchKernelGGL(floor_kernel, dim3(blocks), dim3(256), 0, 0, iters / 4, kind, out);     // warm-up: lets the clock settle
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(floor_kernel, dim3(blocks), dim3(256), 0, 0, iters, kind, out);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double per_simd = (double)iters * 6 * NSET * wps;
            const double ns = ms * 1e6 / per_simd;
            printf("%-28s waves/SIMD %d : %6.2f ns per MFMA per SIMD  (%5.1f cycles @2.4 GHz; %6.0f TFLOP/s executed, %5.0f fp32-equivalent at 3 products)\n",
                   names[kind], wps, ns, ns * 2.4, 32768.0 * 1024 / ns * 1e-3, 32768.0 * 1024 / ns * 1e-3 / 3);
        }
    return 0;
}
