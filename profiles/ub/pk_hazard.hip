// Stand-alone reproducer (no torch) for profiles/r02_packed_fp32_hazard.md: packed-fp32 results of one wave going wrong while a wave
// of another kernel shares its SIMD.  Two HIP streams, one process.
//
//   victims     V_s<FILL>   the exact instruction sequence the compiler emits for the loop body of pw_small_c_forward_kernel
//                           (ag_conv_pointwise.hip built WITH packed fp32): weights through ds_read2_b32, v_pk_mul_f32 x 2,
//                           <FILL>, v_pk_fma_f32 x 6, spelled in inline asm so the form is fixed; every result is compared IN THE KERNEL
//                           with the same arithmetic in scalar v_mul_f32 / v_fma_f32 (bit-identical by IEEE) and mismatches are
//                           counted per (lane quarter, float4 component).  FILL = what sits between the v_pk_mul_f32 that writes the
//                           second register pair and the v_pk_fma_f32 that reads it as src2:
//                             0 `s_nop 0` (what the compiler inserts)   1 `s_nop 1`   2 `s_nop 7`   3 `v_nop`   4 nothing
//                           and two re-selections of the weight halves: 5 every step takes the LOW register of the weight pair for
//                           both results (op_sel_hi:[1,0,..]), 6 every step takes the HIGH register (op_sel:[0,1,..])
//               V_real      pw_small_c_forward_kernel itself (this file includes ag_conv_pointwise.hip and is compiled with packed
//                           fp32 on), 3 -> 512 @32x32 as in profiles/conv_concurrency_pattern.py, bit-compared with its serial result.
//   aggressors  A_<kind>    synthetic: waves issuing ONE kind of instruction back to back (v_cvt_pk_bf16_f32, v_fma_f32, v_pk_fma_f32,
//                           integer VALU, v_cndmask, bf16 MFMA on varying data, LDS read/write stream, global loads), 2 waves per SIMD
//               R_<lib>     the real gather_conv_split_kernel: ag_conv_forward (256 -> 256, 3x3, 128x128) of a libag_hip build, dlopen'ed:
//                           the product library in split_bf16 and in fp32-MFMA arithmetic, and builds of ag_conv.hip with one phase
//                           knocked out (AG_CONV_KNOCKOUT: profiles/ub/build_pk_hazard.sh)
//
// Build + run: profiles/ub/build_pk_hazard.sh && profiles/ub/pk_hazard [reps] [records.jsonl]   (one table; the optional file gets the operands
// of the first wrong results, analysed by profiles/pk_hazard_records.py)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../animatablegaussians_amd/csrc/ag_conv_pointwise.hip"   // the real victim, compiled here with packed fp32 ON

namespace ag {   // the two helpers the pointwise file expects from ag_abi.hip
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int check_hip(hipError_t e, const char* what) { if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return 1; } return 0; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

// counters: [16] mismatches by (lane quarter * 4 + component), [16] = checks
struct BadRec { float x[4][4]; float w[4]; float got[4], want[4]; uint32_t lane, comp, block, iter; };   // x[k][c]: operand k of component c
constexpr int kMaxRec = 48;
struct Counters { unsigned long long bad[16]; unsigned long long checks; unsigned long long first_bad_word[4]; unsigned int nrec; unsigned int pad; BadRec rec[kMaxRec]; };

__device__ __forceinline__ float unit_float(uint32_t h) { return __uint_as_float(0x3f800000u | (h & 0x7fffffu)) - 0.5f; }
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// ---- synthetic victim ----------------------------------------------------------------------------------------------------------
// S0..S3: the op_sel / op_sel_hi suffix of the four steps (which half of the weight pair feeds the low / high result)
#define PK_SEQ(FILLSTR, S0, S1, S2, S3)                                                           \
    asm volatile(                                                                                  \
        "ds_read2_b32 %2, %6 offset1:1\n\t"                                                      \
        "ds_read2_b32 %3, %6 offset0:2 offset1:3\n\t"                                            \
        "s_waitcnt lgkmcnt(1)\n\t"                                                               \
        "v_pk_mul_f32 %0, %8, %2 " S0 "\n\t"               /* A  = x1.lo * w[step 0]      */     \
        "v_pk_mul_f32 %1, %9, %2 " S0 "\n\t"               /* B  = x1.hi * w[step 0]      */     \
        FILLSTR                                                                                    \
        "v_pk_fma_f32 %1, %5, %2, %1 " S1 "\n\t"           /* B += x0.hi * w[step 1]      */     \
        "v_pk_fma_f32 %0, %4, %2, %0 " S1 "\n\t"           /* A += x0.lo * w[step 1]      */     \
        "s_waitcnt lgkmcnt(0)\n\t"                                                               \
        "v_pk_fma_f32 %1, %11, %3, %1 " S2 "\n\t"          /* B += x2.hi * w[step 2]      */     \
        "v_pk_fma_f32 %0, %10, %3, %0 " S2 "\n\t"          /* A += x2.lo * w[step 2]      */     \
        "v_pk_fma_f32 %1, %13, %3, %1 " S3 "\n\t"          /* B += x3.hi * w[step 3]      */     \
        "v_pk_fma_f32 %0, %12, %3, %0 " S3 "\n\t"          /* A += x3.lo * w[step 3]      */     \
        : "=&v"(A), "=&v"(B), "=&v"(w01), "=&v"(w23)                                              \
        : "v"(x0lo), "v"(x0hi), "v"(lds_addr), "v"(0), "v"(x1lo), "v"(x1hi), "v"(x2lo), "v"(x2hi), "v"(x3lo), "v"(x3hi)  \
        : "memory")
// the compiler's forms for pw_small_c_forward_kernel: low result <- HIGH register of the weight pair in steps 0 and 3
#define PK_SEQ_REAL(FILLSTR) PK_SEQ(FILLSTR, "op_sel:[0,1]", "op_sel_hi:[1,0,1]", "op_sel_hi:[1,0,1]", "op_sel:[0,1,0]")

template <int FILL>
__global__ void __launch_bounds__(256) victim_seq_kernel(int iters, uint32_t seed, Counters* out)
{
    __shared__ float sw[64 * 4];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 64 * 4; i += 256) sw[i] = unit_float(mix(seed + 7919u * i + blockIdx.x));
    __syncthreads();
    unsigned bad[4] = { 0, 0, 0, 0 };
    unsigned long long first = 0;
    uint32_t h = mix(seed ^ (blockIdx.x * 256u + tid) * 2654435761u);
    for (int it = 0; it < iters; it++) {
        f2 x0lo, x0hi, x1lo, x1hi, x2lo, x2hi, x3lo, x3hi;
#define GEN(v) h = mix(h + 0x9e3779b9u); v.x = unit_float(h); v.y = unit_float(h >> 7 ^ h << 11);
        GEN(x0lo) GEN(x0hi) GEN(x1lo) GEN(x1hi) GEN(x2lo) GEN(x2hi) GEN(x3lo) GEN(x3hi)
#undef GEN
        const int m = it & 63;
        const int lds_addr = (int)(uintptr_t)(sw + 4 * m) & 0xffff;   // LDS byte address (the compiler materialises it in a VGPR too)
        f2 A, B, w01, w23;
        if constexpr (FILL == 0) PK_SEQ_REAL("s_nop 0\n\t");
        else if constexpr (FILL == 1) PK_SEQ_REAL("s_nop 1\n\t");
        else if constexpr (FILL == 2) PK_SEQ_REAL("s_nop 7\n\t");
        else if constexpr (FILL == 3) PK_SEQ_REAL("v_nop\n\t");
        else if constexpr (FILL == 4) PK_SEQ_REAL("");
        else if constexpr (FILL == 5)   // every step: both results take the LOW register of the weight pair
            PK_SEQ("s_nop 0\n\t", "op_sel_hi:[1,0]", "op_sel_hi:[1,0,1]", "op_sel_hi:[1,0,1]", "op_sel_hi:[1,0,1]");
        else                            // every step: both results take the HIGH register of the weight pair
            PK_SEQ("s_nop 0\n\t", "op_sel:[0,1]", "op_sel:[0,1,0]", "op_sel:[0,1,0]", "op_sel:[0,1,0]");
        // the same arithmetic in scalar instructions (forms fixed in asm as well); (wy, wx, wz, ww) = the weights of steps 0..3... in
        // the order the compiler's forms use them
        const float s0 = sw[4 * m], s1 = sw[4 * m + 1], s2 = sw[4 * m + 2], s3 = sw[4 * m + 3];
        const float wy = FILL == 5 ? s0 : s1, wx = FILL == 6 ? s1 : s0, wz = FILL == 6 ? s3 : s2, ww = FILL == 5 ? s2 : s3;
        float r[4];
        const float a0[4] = { x0lo.x, x0lo.y, x0hi.x, x0hi.y }, a1[4] = { x1lo.x, x1lo.y, x1hi.x, x1hi.y };
        const float a2[4] = { x2lo.x, x2lo.y, x2hi.x, x2hi.y }, a3[4] = { x3lo.x, x3lo.y, x3hi.x, x3hi.y };
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float t;
            asm volatile("v_mul_f32 %0, %1, %2\n\t"
                         "v_fma_f32 %0, %3, %4, %0\n\t"
                         "v_fma_f32 %0, %5, %6, %0\n\t"
                         "v_fma_f32 %0, %7, %8, %0\n\t" : "=&v"(t) : "v"(a1[c]), "v"(wy), "v"(a0[c]), "v"(wx), "v"(a2[c]), "v"(wz), "v"(a3[c]), "v"(ww));
            r[c] = t;
        }
        const float got[4] = { A.x, A.y, B.x, B.y };
        bool any_bad = false;
#pragma unroll
        for (int c = 0; c < 4; c++)
            if (__float_as_uint(got[c]) != __float_as_uint(r[c])) {
                bad[c]++;
                any_bad = true;
                if (!first) first = ((unsigned long long)__float_as_uint(got[c]) << 32) | __float_as_uint(r[c]);
            }
        if (any_bad) {
            const unsigned slot = atomicAdd(&out->nrec, 1u);
            if (slot < (unsigned)kMaxRec) {
                BadRec& br = out->rec[slot];
                for (int c = 0; c < 4; c++) { br.x[0][c] = a0[c]; br.x[1][c] = a1[c]; br.x[2][c] = a2[c]; br.x[3][c] = a3[c]; br.got[c] = got[c]; br.want[c] = r[c]; }
                br.w[0] = wx; br.w[1] = wy; br.w[2] = wz; br.w[3] = ww;
                br.lane = lane; br.comp = 0; br.block = blockIdx.x; br.iter = it;
            }
        }
    }
    const int q = lane >> 4;
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (bad[c]) atomicAdd(&out->bad[q * 4 + c], (unsigned long long)bad[c]);
    if (first) { out->first_bad_word[0] = first; out->first_bad_word[1] = ((unsigned long long)blockIdx.x << 32) | tid; }
    if (tid == 0) atomicAdd(&out->checks, (unsigned long long)iters * 256ull * 4ull);
}

// ---- real victim: compare with the serial result ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) compare_kernel(const float* y, const float* y0, int M, int N, Counters* out)
{
    const int n4 = blockIdx.x * 256 + threadIdx.x;          // the victim's thread index along pixels (4 pixels each)
    if (n4 * 4 >= N) return;
    const int q = (threadIdx.x & 63) >> 4;
    for (int m = blockIdx.y; m < M; m += gridDim.y)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const size_t i = (size_t)m * N + n4 * 4 + c;
            if (__float_as_uint(y[i]) != __float_as_uint(y0[i])) {
                atomicAdd(&out->bad[q * 4 + c], 1ull);
                out->first_bad_word[0] = ((unsigned long long)__float_as_uint(y[i]) << 32) | __float_as_uint(y0[i]);
                out->first_bad_word[1] = ((unsigned long long)m << 32) | (unsigned)(n4 * 4 + c);
            }
        }
    if (threadIdx.x == 0 && blockIdx.y == 0) atomicAdd(&out->checks, (unsigned long long)min(1024, N - blockIdx.x * 1024) * M);
}

// ---- synthetic aggressors ------------------------------------------------------------------------------------------------------------
enum AggKind { AGG_NONE = 0, AGG_CVT_PK_BF16, AGG_FMA, AGG_PK_FMA, AGG_INT, AGG_CNDMASK, AGG_MFMA_BF16, AGG_LDS, AGG_GLOAD, AGG_SPLIT_MIX,
               AGG_COMBO_BF16, AGG_COMBO_F32, AGG_COMBO_BF16_NOGLOAD, AGG_COMBO_BF16_NOLDS, AGG_COMBO_BF16_X4, AGG_KINDS };
static const char* kAggNames[AGG_KINDS] = { "none", "v_cvt_pk_bf16_f32", "v_fma_f32", "v_pk_fma_f32", "v_and/v_lshl (int)", "v_cndmask_b32",
                                           "mfma_32x32x16_bf16 (varying data)", "ds_read_b128+ds_write_b64", "global_load_dword",
                                           "cvt_pk+sub+and+lshl (split_pair mix)",
                                           "COMBO gload + ds_read_b128 + mfma bf16 32x32x16", "COMBO gload + ds_read_b128 + mfma f32 32x32x2",
                                           "COMBO ds_read_b128 + mfma bf16 (no gload)", "COMBO gload + mfma bf16 (no LDS reads)",
                                           "COMBO gload + ds_read + mfma bf16 16x16x32" };

template <int KIND>
__global__ void __launch_bounds__(256) aggressor_kernel(int iters, const float* gmem, float* sink)
{
    __shared__ __attribute__((aligned(16))) float lds[256 * 4 + 64];
    const int tid = threadIdx.x;
    uint32_t h = mix(tid * 2654435761u + blockIdx.x);
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { h = mix(h + i); a[i] = unit_float(h); b[i] = unit_float(h >> 5); }
    lds[tid * 4] = a[0]; lds[tid * 4 + 1] = a[1]; lds[tid * 4 + 2] = a[2]; lds[tid * 4 + 3] = a[3];
    __syncthreads();
    if constexpr (KIND == AGG_MFMA_BF16) {
        f16v acc[2];
        for (int i = 0; i < 16; i++) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
        bf8 x, y;
        for (int i = 0; i < 8; i++) { h = mix(h + i); x[i] = (__bf16)unit_float(h); y[i] = (__bf16)(unit_float(h >> 3) - 1.0f); }
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, acc[1], 0, 0, 0);
            }
        }
        if (acc[0][0] + acc[1][3] == 12345.f) sink[0] = acc[0][1];
        return;
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if constexpr (KIND == AGG_CVT_PK_BF16) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(b[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 3) & 7]));
            } else if constexpr (KIND == AGG_FMA) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
            } else if constexpr (KIND == AGG_PK_FMA) {
                f2* ap = reinterpret_cast<f2*>(a); f2* bp = reinterpret_cast<f2*>(b);
#pragma unroll
                for (int i = 0; i < 4; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(ap[i]) : "v"(bp[i]), "v"(bp[(i + 1) & 3]));
            } else if constexpr (KIND == AGG_INT) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_and_b32 %0, %1, %0\n\tv_lshlrev_b32 %0, 1, %0" : "+v"(a[i]) : "v"(b[i]));
            } else if constexpr (KIND == AGG_CNDMASK) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]));
            } else if constexpr (KIND == AGG_LDS) {
                f4 v; f2 w = { a[0], a[1] };
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((tid * 16) & 0xfff) : "memory");
                asm volatile("ds_write_b64 %0, %1" :: "v"((tid * 8) & 0xfff), "v"(w) : "memory");
                a[0] += v[0]; a[1] += v[3];
            } else if constexpr (KIND == AGG_GLOAD) {
#pragma unroll
                for (int i = 0; i < 4; i++) a[i] += gmem[((h >> (i + 3)) + it * 64 + u) & 0xfffff];
            } else if constexpr (KIND == AGG_SPLIT_MIX) {
                // the loader arithmetic of gather_conv_split_kernel (split_pair), instruction forms fixed
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    uint32_t p0, p1; float r0, r1;
                    asm volatile("v_cvt_pk_bf16_f32 %0, %4, %5\n\t"
                                 "v_lshlrev_b32 %2, 16, %0\n\t"
                                 "v_and_b32 %3, 0xffff0000, %0\n\t"
                                 "v_sub_f32 %2, %4, %2\n\t"
                                 "v_sub_f32 %3, %5, %3\n\t"
                                 "v_cvt_pk_bf16_f32 %1, %2, %3\n\t"
                                 : "=&v"(p0), "=&v"(p1), "=&v"(r0), "=&v"(r1) : "v"(a[i]), "v"(a[i + 1]));
                    a[i] = r0 + __uint_as_float(p1 << 16); a[i + 1] = r1 + __uint_as_float(p0 & 0x7fff0000u);
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i] + b[i];
    if (s == 12345.f) sink[0] = s;
}

// The three phases the knock-outs of the real kernel showed to be necessary TOGETHER, in one synthetic kernel: global loads in flight,
// ds_read_b128 operand reads, matrix instructions.  Compiler-generated (builtins), as in the real kernel, so every intra-wave
// hazard is the compiler's business.
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ void __launch_bounds__(512, 4) combo_aggressor_kernel(int iters, const float* __restrict__ gmem, float* sink)
{
    constexpr bool GL = KIND != AGG_COMBO_BF16_NOGLOAD, LD = KIND != AGG_COMBO_BF16_NOLDS;
    __shared__ __attribute__((aligned(16))) char lds[48 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 48 * 1024 / 4; i += 512) reinterpret_cast<float*>(lds)[i] = unit_float(mix(i + blockIdx.x));
    __syncthreads();
    f16v acc[2][2];
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int i = 0; i < 16; i++) acc[a][b][i] = 0.f;
    float stage[8];
    for (int j = 0; j < 8; j++) stage[j] = 0.f;
    uint32_t off = (blockIdx.x * 512u + tid) * 4u;
    float keep = 0.f;
    for (int it = 0; it < iters; it++) {
        // operand reads of this tile
        bf8v A[2][3], B[2][3];
        const int base = ((it & 1) * 24 * 1024) + (lane & 31) * 32 + (lane >> 5) * 16 + wave * 1024;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int pl = 0; pl < 3; pl++) {
                if constexpr (LD) {
                    A[i][pl] = *reinterpret_cast<const bf8v*>(lds + ((base + (i * 3 + pl) * 2048) & (48 * 1024 - 16)));
                    B[i][pl] = *reinterpret_cast<const bf8v*>(lds + ((base + (i * 3 + pl) * 2048 + 12288) & (48 * 1024 - 16)));
                } else {
                    f32x4v t = { stage[0] + i, stage[1] + pl, 1.f, 2.f };
                    asm volatile("" : "+v"(t));
                    A[i][pl] = __builtin_bit_cast(bf8v, t); B[i][pl] = __builtin_bit_cast(bf8v, t);
                }
            }
        // gathers of the tile after next (consumed two iterations later through `keep`)
        if constexpr (GL) {
#pragma unroll
            for (int j = 0; j < 8; j++) { keep += stage[j]; stage[j] = gmem[((off >> 2) + j * 65536u + it * 512u) & 0xfffffu]; }
        }
        off += 2048u;
        constexpr int ta[6] = { 2, 1, 0, 1, 0, 0 }, tb[6] = { 0, 1, 2, 0, 1, 0 };
#pragma unroll
        for (int t = 0; t < 6; t++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                if constexpr (KIND == AGG_COMBO_F32) {
                    const f32x4v af = __builtin_bit_cast(f32x4v, A[i][ta[t]]), bf = __builtin_bit_cast(f32x4v, B[0][tb[t]]);
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc[i][0], 0, 0, 0);
                } else if constexpr (KIND == AGG_COMBO_BF16_X4) {
                    typedef float f4a __attribute__((ext_vector_type(4)));
                    f4a c4 = { acc[i][0][0], acc[i][0][1], acc[i][0][2], acc[i][0][3] };
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i][ta[t]], B[0][tb[t]], c4, 0, 0, 0);
                    acc[i][0][0] = c4[0]; acc[i][0][1] = c4[1]; acc[i][0][2] = c4[2]; acc[i][0][3] = c4[3];
                } else {
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][ta[t]], B[0][tb[t]], acc[i][0], 0, 0, 0);
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    float s = keep;
    for (int j = 0; j < 8; j++) s += stage[j];
    for (int a = 0; a < 2; a++) for (int i = 0; i < 16; i++) s += acc[a][0][i];
    if (s == 12345.f) sink[0] = s;
}

// ---- real aggressor: a libag_hip build --------------------------------------------------------------------------------------------------
struct AgConvDescC { int32_t kind, Cin, Cout, H, W, k, stride, padding; float weight_scale; };
struct ConvLib {
    std::string name;
    void* h = nullptr;
    int (*forward)(const AgConvDescC*, const float*, const float*, const float*, const float*, float*, void*, size_t, void*) = nullptr;
    size_t (*ws_bytes)(const AgConvDescC*) = nullptr;
    int (*set_math)(int) = nullptr;
    int math = 1;
};

static bool load_lib(ConvLib& L, const std::string& path, const std::string& name, int math)
{
    L.h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!L.h) { fprintf(stderr, "  (skipping %s: %s)\n", name.c_str(), dlerror()); return false; }
    L.forward = reinterpret_cast<decltype(L.forward)>(dlsym(L.h, "ag_conv_forward"));
    L.ws_bytes = reinterpret_cast<decltype(L.ws_bytes)>(dlsym(L.h, "ag_conv_workspace_bytes"));
    L.set_math = reinterpret_cast<decltype(L.set_math)>(dlsym(L.h, "ag_conv_set_math"));
    L.name = name; L.math = math;
    return L.forward && L.ws_bytes && L.set_math;
}

struct Cell { unsigned long long bad[16]; unsigned long long checks; unsigned long long first[2]; };

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 6;
    std::string here = argv[0];
    here = here.substr(0, here.find_last_of('/') == std::string::npos ? 0 : here.find_last_of('/'));
    if (here.empty()) here = ".";
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s (%s), %d CUs\n", prop.name, prop.gcnArchName, prop.multiProcessorCount);

    hipStream_t sa, sv; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    Counters* dcnt; CK(hipMalloc(&dcnt, sizeof(Counters)));
    float* sink; CK(hipMalloc(&sink, 64));
    float* gmem; CK(hipMalloc(&gmem, (1u << 20) * 4 + 4096)); CK(hipMemset(gmem, 0, (1u << 20) * 4 + 4096));

    // real victim buffers: 3 -> 512 @32x32 (profiles/conv_concurrency_pattern.py) and 3 -> 128 @256x256 (the network's first FromRGB)
    struct RealVictim { int C, M, H; float *x, *w, *y, *y0; AgConvDesc d; };
    std::vector<RealVictim> rv = { { 3, 512, 32 }, { 3, 128, 256 } };
    srand(1234);
    for (auto& v : rv) {
        const size_t N = (size_t)v.H * v.H;
        std::vector<float> hx(v.C * N), hw((size_t)v.M * v.C);
        for (auto& f : hx) f = (float)rand() / RAND_MAX * 2.f - 1.f;
        for (auto& f : hw) f = (float)rand() / RAND_MAX * 2.f - 1.f;
        CK(hipMalloc(&v.x, hx.size() * 4)); CK(hipMalloc(&v.w, hw.size() * 4)); CK(hipMalloc(&v.y, v.M * N * 4)); CK(hipMalloc(&v.y0, v.M * N * 4));
        CK(hipMemcpy(v.x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(v.w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        v.d = AgConvDesc{ AG_CONV, v.C, v.M, v.H, v.H, 1, 1, 0, 1.0f };
        if (ag::pointwise_forward(&v.d, v.x, v.w, nullptr, nullptr, v.y0, sv) != 1) { fprintf(stderr, "pointwise_forward refused\n"); return 2; }
        CK(hipStreamSynchronize(sv));
    }

    // real aggressors
    std::vector<ConvLib> libs;
    struct { const char* file; const char* name; int math; } wanted[] = {
        { "/../../animatablegaussians_amd/lib/libag_hip.so", "R product split_bf16", 1 },
        { "/../../animatablegaussians_amd/lib/libag_hip.so", "R product fp32-MFMA", 0 },
        { "/ko/libag_ko1.so", "R split, no global gathers", 1 },
        { "/ko/libag_ko2.so", "R split, no split arithmetic (no cvt_pk)", 1 },
        { "/ko/libag_ko4.so", "R split, no LDS writes", 1 },
        { "/ko/libag_ko8.so", "R split, no LDS operand reads", 1 },
        { "/ko/libag_ko16.so", "R split, no MFMA", 1 },
        { "/ko/libag_ko24.so", "R split, loader only (no LDS reads, no MFMA)", 1 },
        { "/ko/libag_ko7.so", "R split, LDS reads + MFMA only", 1 },
        { "/ko/libag_ko15.so", "R split, MFMA only (+barriers)", 1 },
        { "/ko/libag_ko22.so", "R split, gathers + LDS reads only", 1 },
    };
    for (auto& w : wanted) { ConvLib L; if (load_lib(L, here + w.file, w.name, w.math)) libs.push_back(L); }
    AgConvDescC cd{ 0, 256, 256, 128, 128, 3, 1, 1, 1.0f };
    float *cx, *cw, *cy; void* cws; size_t cws_bytes = 0;
    {
        std::vector<float> hx((size_t)256 * 128 * 128), hw((size_t)256 * 256 * 9);
        for (auto& f : hx) f = (float)rand() / RAND_MAX * 2.f - 1.f;
        for (auto& f : hw) f = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
        CK(hipMalloc(&cx, hx.size() * 4)); CK(hipMalloc(&cw, hw.size() * 4)); CK(hipMalloc(&cy, hx.size() * 4));
        CK(hipMemcpy(cx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(cw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        for (auto& L : libs) cws_bytes = std::max(cws_bytes, L.ws_bytes(&cd));
        CK(hipMalloc(&cws, cws_bytes + 4096));
    }

    const int n_syn = AGG_KINDS, n_agg = n_syn + (int)libs.size();
    constexpr int kSynVictims = 7;
    const int n_vic = kSynVictims + (int)rv.size();
    static const char* vic_names[] = { "V_s s_nop 0", "V_s s_nop 1", "V_s s_nop 7", "V_s v_nop", "V_s no fill", "V_s all low-sel", "V_s all high-sel", "V_real 3->512@32", "V_real 3->128@256" };
    std::vector<Cell> cells((size_t)n_agg * n_vic);
    memset(cells.data(), 0, cells.size() * sizeof(Cell));

    auto launch_aggressor = [&](int a) {
        const int blocks = prop.multiProcessorCount * 2, iters = 60000;
#define LA(K) case K: hipLaunchKernelGGL(aggressor_kernel<K>, dim3(blocks), dim3(256), 0, sa, K == AGG_MFMA_BF16 ? iters / 8 : (K == AGG_GLOAD || K == AGG_LDS) ? iters / 16 : iters, gmem, sink); break
        if (a < n_syn) {
            switch (a) { case AGG_NONE: break; LA(AGG_CVT_PK_BF16); LA(AGG_FMA); LA(AGG_PK_FMA); LA(AGG_INT); LA(AGG_CNDMASK); LA(AGG_MFMA_BF16); LA(AGG_LDS); LA(AGG_GLOAD); LA(AGG_SPLIT_MIX);
#define LC(K) case K: hipLaunchKernelGGL(combo_aggressor_kernel<K>, dim3(prop.multiProcessorCount * 2), dim3(512), 0, sa, 2500, gmem, sink); break
            LC(AGG_COMBO_BF16); LC(AGG_COMBO_F32); LC(AGG_COMBO_BF16_NOGLOAD); LC(AGG_COMBO_BF16_NOLDS); LC(AGG_COMBO_BF16_X4);
#undef LC
            }
        } else {
            ConvLib& L = libs[a - n_syn];
            L.set_math(L.math);
            for (int i = 0; i < 24; i++) L.forward(&cd, cx, cw, nullptr, nullptr, cy, cws, cws_bytes + 4096, sa);
        }
#undef LA
    };
    auto launch_victim = [&](int v, uint32_t seed) {
        const int blocks = 512, iters = 400;
        switch (v) {
        case 0: hipLaunchKernelGGL(victim_seq_kernel<0>, dim3(blocks), dim3(256), 0, sv, iters, seed, dcnt); break;
        case 1: hipLaunchKernelGGL(victim_seq_kernel<1>, dim3(blocks), dim3(256), 0, sv, iters, seed, dcnt); break;
        case 2: hipLaunchKernelGGL(victim_seq_kernel<2>, dim3(blocks), dim3(256), 0, sv, iters, seed, dcnt); break;
        case 3: hipLaunchKernelGGL(victim_seq_kernel<3>, dim3(blocks), dim3(256), 0, sv, iters, seed, dcnt); break;
        case 4: hipLaunchKernelGGL(victim_seq_kernel<4>, dim3(blocks), dim3(256), 0, sv, iters, seed, dcnt); break;
        case 5: hipLaunchKernelGGL(victim_seq_kernel<5>, dim3(blocks), dim3(256), 0, sv, iters, seed, dcnt); break;
        case 6: hipLaunchKernelGGL(victim_seq_kernel<6>, dim3(blocks), dim3(256), 0, sv, iters, seed, dcnt); break;
        default: {
            RealVictim& r = rv[v - kSynVictims];
            const int N = r.H * r.H;
            CK(hipMemsetAsync(r.y, 0, (size_t)r.M * N * 4, sv));
            ag::pointwise_forward(&r.d, r.x, r.w, nullptr, nullptr, r.y, sv);
            hipLaunchKernelGGL(compare_kernel, dim3((N / 4 + 255) / 256, 16), dim3(256), 0, sv, r.y, r.y0, r.M, N, dcnt);
        } }
    };

    FILE* recf = fopen(argc > 2 ? argv[2] : "/dev/null", "w");
    int records_printed = 0;
    for (int rep = 0; rep < reps; rep++)
        for (int a = 0; a < n_agg; a++)
            for (int v = 0; v < n_vic; v++) {
                CK(hipMemset(dcnt, 0, sizeof(Counters)));
                CK(hipDeviceSynchronize());
                launch_aggressor(a);
                const int vl = v < kSynVictims ? 12 : 40;
                for (int i = 0; i < vl; i++) launch_victim(v, 1000u * rep + 17u * i + 3u);
                CK(hipDeviceSynchronize());
                Counters hc; CK(hipMemcpy(&hc, dcnt, sizeof(hc), hipMemcpyDeviceToHost));
                Cell& c = cells[(size_t)a * n_vic + v];
                for (int i = 0; i < 16; i++) c.bad[i] += hc.bad[i];
                c.checks += hc.checks;
                if (hc.first_bad_word[0] && !c.first[0]) { c.first[0] = hc.first_bad_word[0]; c.first[1] = hc.first_bad_word[1]; }
                if (v < kSynVictims && hc.nrec && records_printed < 400 && (v == 0 || v >= 5)) {
                    const unsigned nr = hc.nrec < (unsigned)kMaxRec ? hc.nrec : (unsigned)kMaxRec;
                    for (unsigned i = 0; i < nr && records_printed < 400; i++, records_printed++) {
                        const BadRec& b = hc.rec[i];
                        fprintf(recf, "{\"agg\": \"%s\", \"fill\": %d, \"lane\": %u, \"block\": %u, \"iter\": %u, \"w\": [%a, %a, %a, %a], ",
                                a < n_syn ? kAggNames[a] : libs[a - n_syn].name.c_str(), v, b.lane, b.block, b.iter, b.w[0], b.w[1], b.w[2], b.w[3]);
                        fprintf(recf, "\"x\": [");
                        for (int k = 0; k < 4; k++) fprintf(recf, "[%a, %a, %a, %a]%s", b.x[k][0], b.x[k][1], b.x[k][2], b.x[k][3], k < 3 ? ", " : "");
                        fprintf(recf, "], \"got\": [%a, %a, %a, %a], \"want\": [%a, %a, %a, %a]}\n", b.got[0], b.got[1], b.got[2], b.got[3], b.want[0], b.want[1], b.want[2], b.want[3]);
                    }
                }
            }

    // time overlap sanity: how long one aggressor launch sequence and one victim sequence take alone
    printf("\nmismatching results / checked results, %d repetitions per cell (every victim result is checked bit for bit)\n", reps);
    printf("%-46s", "aggressor \\ victim");
    for (int v = 0; v < n_vic; v++) printf(" | %-17s", vic_names[v]);
    printf("\n");
    for (int a = 0; a < n_agg; a++) {
        printf("%-46s", a < n_syn ? kAggNames[a] : libs[a - n_syn].name.c_str());
        for (int v = 0; v < n_vic; v++) {
            const Cell& c = cells[(size_t)a * n_vic + v];
            unsigned long long tot = 0;
            for (int i = 0; i < 16; i++) tot += c.bad[i];
            char buf[64]; snprintf(buf, sizeof buf, "%llu/%.1e", tot, (double)c.checks);
            printf(" | %-17s", buf);
        }
        printf("\n");
    }
    if (recf) fclose(recf);
    printf("\nwhere the mismatches sit (rows: lanes 0-15 / 16-31 / 32-47 / 48-63; columns: float4 component 0..3), cells with any:\n");
    for (int a = 0; a < n_agg; a++)
        for (int v = 0; v < n_vic; v++) {
            const Cell& c = cells[(size_t)a * n_vic + v];
            unsigned long long tot = 0;
            for (int i = 0; i < 16; i++) tot += c.bad[i];
            if (!tot) continue;
            printf("  [%s] x [%s]:", a < n_syn ? kAggNames[a] : libs[a - n_syn].name.c_str(), vic_names[v]);
            for (int q = 0; q < 4; q++) printf("  q%d: %llu %llu %llu %llu", q, c.bad[q * 4], c.bad[q * 4 + 1], c.bad[q * 4 + 2], c.bad[q * 4 + 3]);
            printf("   e.g. got %08x want %08x (id %llx)\n", (unsigned)(c.first[0] >> 32), (unsigned)c.first[0], c.first[1]);
        }
    return 0;
}
