// Microbenchmark (round 6): issue rate of the cross-lane instructions the blend kernels' reductions are made of, against v_add_f32.
//   hipcc --offload-arch=gfx950 -O3 -o xlane_rate xlane_rate.hip && ./xlane_rate        -> cycles per wave instruction per SIMD at 4 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>

#define BODY8(INSTR)                                                                                                               \
    asm volatile(INSTR(0, 1) INSTR(2, 3) INSTR(4, 5) INSTR(6, 7) INSTR(0, 1) INSTR(2, 3) INSTR(4, 5) INSTR(6, 7)                    \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));

#define I_ADD(A, B) "v_add_f32 %" #A ", %" #A ", %" #B "\n"
#define I_SWAP32(A, B) "v_permlane32_swap_b32 %" #A ", %" #B "\n"
#define I_SWAP16(A, B) "v_permlane16_swap_b32 %" #A ", %" #B "\n"
#define I_DPPQ(A, B) "v_add_f32_dpp %" #A ", %" #B ", %" #A " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_DPPR(A, B) "v_add_f32_dpp %" #A ", %" #B ", %" #A " row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_MOVDPP(A, B) "v_mov_b32_dpp %" #A ", %" #B " row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_SWZ(A, B) "ds_swizzle_b32 %" #A ", %" #B " offset:swizzle(BITMASK_PERM, \"01pip\")\n s_waitcnt lgkmcnt(0)\n"
#define I_EXP(A, B) "v_exp_f32 %" #A ", %" #B "\n"
#define I_RCP(A, B) "v_rcp_f32 %" #A ", %" #B "\n"
#define I_FMA(A, B) "v_fma_f32 %" #A ", %" #A ", %" #B ", %" #B "\n"
#define I_CNDMASK(A, B) "v_cndmask_b32 %" #A ", %" #A ", %" #B ", vcc\n"
#define I_MED3(A, B) "v_med3_f32 %" #A ", %" #A ", %" #B ", %" #B "\n"
#define I_CNDS(A, B) "v_cndmask_b32_e64 %" #A ", %" #A ", %" #B ", s[20:21]\n"
#define I_CMPCND(A, B) "v_cmp_gt_f32 vcc, %" #A ", %" #B "\n v_cndmask_b32 %" #A ", %" #A ", %" #B ", vcc\n"
#define I_CMPS(A, B) "v_cmp_gt_f32 s[20:21], %" #A ", %" #B "\n"
#define I_MUL(A, B) "v_mul_f32 %" #A ", %" #A ", %" #B "\n"
#define I_MAX(A, B) "v_max_f32 %" #A ", %" #A ", %" #B "\n"
#define I_FMAC(A, B) "v_fmac_f32 %" #A ", %" #B ", %" #B "\n"
#define I_CVT(A, B) "v_cvt_f32_u32 %" #A ", %" #B "\n"
#define I_MOV(A, B) "v_mov_b32 %" #A ", %" #B "\n"
#define I_BPERM(A, B) "ds_bpermute_b32 %" #A ", %" #B ", %" #A "\n"
#define I_READLANE(A, B) "v_readlane_b32 s22, %" #A ", 5\n"

#define KERNEL(NAME, INSTR)                                                                                      \
    __global__ void NAME(int iters, float* out)                                                                  \
    {                                                                                                            \
        float a0 = threadIdx.x, a1 = 1.5f, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;                         \
        for (int i = 0; i < iters; i++) {                                                                        \
            BODY8(INSTR) BODY8(INSTR) BODY8(INSTR) BODY8(INSTR)                                                  \
        }                                                                                                        \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) out[0] = a0;                                       \
    }
KERNEL(k_add, I_ADD) KERNEL(k_swap32, I_SWAP32) KERNEL(k_swap16, I_SWAP16) KERNEL(k_dppq, I_DPPQ) KERNEL(k_dppr, I_DPPR)
KERNEL(k_cnds, I_CNDS) KERNEL(k_cmpcnd, I_CMPCND) KERNEL(k_cmps, I_CMPS) KERNEL(k_mul, I_MUL) KERNEL(k_max, I_MAX) KERNEL(k_fmac, I_FMAC) KERNEL(k_cvt, I_CVT) KERNEL(k_mov, I_MOV) KERNEL(k_readlane, I_READLANE)
KERNEL(k_movdpp, I_MOVDPP) KERNEL(k_exp, I_EXP) KERNEL(k_rcp, I_RCP) KERNEL(k_fma, I_FMA) KERNEL(k_cnd, I_CNDMASK) KERNEL(k_med3, I_MED3)

template <typename K>
void run(const char* name, K kern, float* d)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * 4, thr = 256;       // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(thr), 0, 0, 10, d);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(thr), 0, 0, iters, d);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)blocks * (thr / 64) * iters * 32 / 1024.0;      // wave instructions per SIMD
    printf("%-28s %8.3f ms  %6.2f cycles per wave instruction per SIMD (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / per_simd);
}

int main()
{
    float* d; (void)hipMalloc(&d, 4);
    run("v_add_f32", k_add, d);
    run("v_fma_f32", k_fma, d);
    run("v_permlane32_swap_b32", k_swap32, d);
    run("v_permlane16_swap_b32", k_swap16, d);
    run("v_add_f32_dpp quad_perm", k_dppq, d);
    run("v_add_f32_dpp row_shr:4", k_dppr, d);
    run("v_mov_b32_dpp row_shr:8", k_movdpp, d);
    run("v_exp_f32", k_exp, d);
    run("v_rcp_f32", k_rcp, d);
    run("v_cndmask_b32", k_cnd, d);
    run("v_med3_f32", k_med3, d);
    run("v_cndmask_b32_e64 (sgpr mask)", k_cnds, d);
    run("v_cmp + v_cndmask (2 instr)", k_cmpcnd, d);
    run("v_cmp_gt_f32 -> sgpr pair", k_cmps, d);
    run("v_mul_f32", k_mul, d);
    run("v_max_f32", k_max, d);
    run("v_fmac_f32", k_fmac, d);
    run("v_cvt_f32_u32", k_cvt, d);
    run("v_mov_b32", k_mov, d);
    run("v_readlane_b32", k_readlane, d);
    return 0;
}
