/*
 * ag_conv.h — C ABI of the MFMA convolution kernels for the StyleUNet (libag_hip.so), batch 1, fp32.
 *
 * Boundary B4 of SURVEY.md: replaces the cuDNN calls behind network/styleunet/conv2d_gradfix.py:22-75
 * (conv2d / conv_transpose2d, reached from dual_styleunet.py:114,239-298).  Only the configurations that occur in
 * the product are supported: conv2d k x k (k = 1, 3, 4) with stride 1 or 2 and any zero padding, and
 * conv_transpose2d 3x3 stride 2 padding 0; groups = batch = 1 (the reference's grouped call with groups = batch
 * is the same thing at batch 1).
 *
 * All three GEMM-shaped problems of a convolution run on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32
 * accumulation): forward, input gradient (= a gather-convolution with re-packed weights) and weight gradient.
 * Device pointers, contiguous NCHW without the batch dimension; 0 on success (codes in ag_raster.h).
 */
#ifndef AG_CONV_H
#define AG_CONV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum AgConvKind {
    AG_CONV = 0,            /* y = conv2d(x, w [Cout,Cin,k,k], stride, padding) */
    AG_CONV_TRANSPOSE = 1   /* y = conv_transpose2d(x, w [Cin,Cout,k,k], stride 2, padding 0): OH = (H-1)*2 + k */
} AgConvKind;

typedef struct AgConvDesc {
    int32_t kind;          /* AgConvKind */
    int32_t Cin, Cout;
    int32_t H, W;          /* input spatial size */
    int32_t k;             /* square kernel size */
    int32_t stride;        /* 1 or 2 (AG_CONV_TRANSPOSE: 2) */
    int32_t padding;       /* AG_CONV only */
    float weight_scale;    /* the convolution uses w * weight_scale (EqualConv2d's `self.weight * self.scale`,
                              dual_styleunet.py:100-117: formed as that fp32 product while the weights are re-packed, so the result
                              equals convolving with the pre-scaled tensor); dw is the gradient w.r.t. the UN-scaled w.
                              0 is read as 1 (no scaling) */
} AgConvDesc;

/* Output spatial size of the described convolution. */
int ag_conv_output_size(const AgConvDesc* d, int32_t* OH, int32_t* OW);

/* Bytes of scratch the three entry points need (re-packed weights / transposed operands). */
size_t ag_conv_workspace_bytes(const AgConvDesc* d);

/*
 * y [Cout, OH, OW] = conv(x [Cin, H, W], w) * out_scale[co] + bias[co]
 * out_scale / bias may be NULL (1 / 0).  out_scale is the demodulation coefficient hook of ModulatedConv2d
 * (dual_styleunet.py:246-250); bias the EqualConv2d bias (:114-120).
 */
int ag_conv_forward(const AgConvDesc* d, const float* x, const float* w, const float* out_scale, const float* bias,
                    float* y, void* workspace, size_t workspace_bytes, void* stream);

/* dx [Cin, H, W] = d(loss)/dx given dy [Cout, OH, OW] (dy already multiplied by out_scale if one was used). */
int ag_conv_backward_input(const AgConvDesc* d, const float* dy, const float* w, float* dx, void* workspace,
                           size_t workspace_bytes, void* stream);

/* dw (same shape as w) = d(loss)/dw given x and dy.  dw is overwritten. */
int ag_conv_backward_weight(const AgConvDesc* d, const float* x, const float* dy, float* dw, void* workspace,
                            size_t workspace_bytes, void* stream);

/* Calibration: `blocks` x 4 waves each issue iters x 4 independent v_mfma_f32_32x32x2_f32 (8192 FLOP each per wave) with
 * no memory traffic; time it to get the attainable fp32 MFMA rate of the device (profiles/mfma_peak.py). */
/* Arithmetic of the MFMA convolutions (process-wide; the pointwise VALU kernels are plain fp32 either way).
 *   AG_CONV_MATH_FP32_MFMA     v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation.
 *   AG_CONV_MATH_SPLIT_F16     (default since round 4) every operand tensor scaled by a power of two that puts its largest magnitude M into
 *                              [2^14, 2^15) and every scaled value written as the sum of two fp16 parts (22 significant bits), the three
 *                              products a_h b_h + a_h b_l + a_l b_h on v_mfma_f32_32x32x16_f16, fp32 accumulation, the scales removed in the
 *                              epilogue.  Per output: |error| <= 3 * 2^-24 sum |a| |b| + 2^-39 (M_b sum |a| + M_a sum |b|) -- the grade of
 *                              SPLIT_BF16 for every output whose operands are on average within 2^16 of their tensors' maxima, an absolute
 *                              error 2^-15 of fp32's own rounding on the tensor's large outputs for the others.  Half the matrix
 *                              instructions of SPLIT_BF16; costs one sweep for the maximum of every operand a producer did not hand over.
 *   AG_CONV_MATH_SPLIT_BF16    every fp32 operand split into three bf16 parts (x = x0 + x1 + x2 to 2^-26 |x|), the six
 *                              products a_i b_j with i + j <= 2 on v_mfma_f32_32x32x16_bf16, fp32 accumulation: each product is
 *                              within 2^-23 |a| |b| of the exact one -- the size of fp32's own product rounding.
 *   AG_CONV_MATH_SPLIT_BF16X3  opt-in: the three products with i + j <= 1, half the matrix work: each product within 3 * 2^-16 |a| |b|.
 *                              Not fp32-grade; for scale, the reference's cuDNN path runs TF32 (2^-11 per operand) by default --
 *                              conv2d_gradfix.py never disables it.
 *   AG_CONV_MATH_F16           opt-in (round 5): ONE fp16 part per operand under SPLIT_F16's per-tensor scale, one product per fp32 product on
 *                              v_mfma_f32_32x32x16_f16, fp32 accumulation.  Operands carry 11 significant bits, rounded to nearest: the operand
 *                              grade of cuDNN's TF32 path (10 explicit mantissa bits), i.e. the arithmetic the reference's own convolutions
 *                              run on its hardware (network/styleunet/conv2d_gradfix.py:185-189 passes torch.backends.cudnn.allow_tf32,
 *                              True by default, and nothing clears it).  Contract: the result equals the convolution of the operands
 *                              ROUNDED to 11 significant bits (exact products, fp32 accumulation) wherever |x| >= 2^-29 of its tensor's
 *                              largest magnitude; below that fp16's gradual underflow applies (absolute error <= 2^-40 M per operand).
 *                              A third of SPLIT_F16's matrix instructions.  Not fp32-grade; no fp32 parity claim is made in this mode.
 * Returns AG_OK / AG_ERR_INVALID_ARGUMENT. */
typedef enum AgConvMath { AG_CONV_MATH_FP32_MFMA = 0, AG_CONV_MATH_SPLIT_BF16 = 1, AG_CONV_MATH_SPLIT_BF16X3 = 2, AG_CONV_MATH_SPLIT_F16 = 3, AG_CONV_MATH_F16 = 4 } AgConvMath;
int ag_conv_set_math(int mode);
int ag_conv_get_math(void);

/* Range guard of the scaled fp16 forms (round 5).  SPLIT_F16 / F16 scale every operand tensor by its largest magnitude, which may be HANDED to a
 * call (ag_layers.h x_maxima / operand_maxima, ConvOpts) instead of swept: a maximum that is too small (stale after an in-place write the caller
 * did not account for) overflows fp16 and would silently put inf / NaN into the outputs.  Every MFMA convolution kernel therefore tests its
 * accumulators in the epilogue (32 class tests per lane, once per tile) and raises a sticky flag in host-visible memory when one is not finite
 * (the same happens for operands that were not finite to begin with).  The flag is read WITHOUT synchronising:
 *   - the NEXT convolution call on any stream returns AG_ERR_RANGE (and clears the flag) instead of launching,
 *   - ag_conv_status(clear) returns AG_ERR_RANGE / AG_OK on demand (after a stream synchronisation it covers everything enqueued before).
 * A kernel that raised the flag still wrote its (non-finite) outputs; nothing is repaired.  Round 6: the guard is armed in SPLIT_F16 / F16 only
 * (in the fp32 / bf16 forms a non-finite operand propagates into the outputs as it does through the reference's cuDNN calls, and nothing is
 * refused), and the flag carries which launch raised it (kernel kind, output rows, gathered channels: named in ag_last_error). */
int ag_conv_status(int clear);

int ag_debug_mfma_rate(int blocks, int iters, float* out, void* stream);
/* Same for v_mfma_f32_32x32x16_bf16 (calibration of the bf16-split option; not used by the product). */
int ag_debug_mfma_rate_bf16(int blocks, int iters, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AG_CONV_H */
