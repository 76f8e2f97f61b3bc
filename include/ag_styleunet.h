/*
 * ag_styleunet.h — C ABI of the StyleUNet element-wise / FIR operators (libag_hip.so).
 *
 * Drop-in for the two native extension modules the reference's network/styleunet imports as top-level modules:
 *   `fused`      fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)
 *                network/styleunet/fused_bias_act.cpp:17-31, fused_bias_act_kernel.cu:18-104
 *   `upfirdn2d`  upfirdn2d.upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
 *                network/styleunet/upfirdn2d.cpp:17-31, upfirdn2d_kernel.cu:49-368
 * Device pointers, fp32, contiguous; 0 on success (codes in ag_raster.h).
 */
#ifndef AG_STYLEUNET_H
#define AG_STYLEUNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * out[i] = scale * f(x[i] + bias[(i / step_b) % size_b], ref[i])
 *   act 1 (linear): grad 0/1 -> x            ; grad 2 -> 0
 *   act 3 (lrelu) : grad 0 -> x > 0 ? x : alpha x ; grad 1 -> ref > 0 ? x : alpha x ; grad 2 -> 0
 * bias == NULL or size_b == 0: no bias.  ref == NULL: ref = 0.  step_b = product of the dimensions after dim 1.
 * Any other (act, grad) pair behaves like act 1 / grad 0, as the reference's `default:` label does.
 */
int ag_fused_bias_act(float* out, const float* x, const float* bias, const float* ref, int32_t act, int32_t grad,
                      float alpha, float scale, int64_t size_x, int64_t step_b, int32_t size_b, void* stream);

/*
 * input [major, in_h, in_w] (the reference's [major, in_h, in_w, minor = 1]), kernel [kernel_h, kernel_w]:
 * zero-insert upsampling by (up_y, up_x), padding (negative = cropping), correlation with the FLIPPED kernel,
 * decimation by (down_y, down_x).  out [major, out_h, out_w] with
 *   out_h = (in_h*up_y + pad_y0 + pad_y1 - kernel_h + down_y) / down_y   (same for w).
 * Returns AG_ERR_INVALID_ARGUMENT when the output would be empty.
 */
int ag_upfirdn2d(float* out, const float* input, const float* kernel, int32_t major, int32_t in_h, int32_t in_w,
                 int32_t kernel_h, int32_t kernel_w, int32_t up_x, int32_t up_y, int32_t down_x, int32_t down_y,
                 int32_t pad_x0, int32_t pad_x1, int32_t pad_y0, int32_t pad_y1, void* stream);

/*
 * The tail of StyledConv / ConvLayer in one pass (network/styleunet/dual_styleunet.py:303-313,598-604 = NoiseInjection
 * followed by FusedLeakyReLU; :367-369 without noise), batch 1:
 *   y[c][p] = lrelu(x[c][p] + noise_weight[0] * noise[p] + bias[c], slope) * scale        x, y [C][HW], noise [HW]
 * noise == NULL: no noise term; bias == NULL: no bias.  Same arithmetic as `image + weight * noise` followed by
 * fused_bias_act(act 3, grad 0).
 */
int ag_noise_bias_act_forward(float* y, const float* x, const float* noise, const float* noise_weight, const float* bias,
                              int32_t C, int32_t HW, float slope, float scale, void* stream);

/*
 * Backward of the above from the saved OUTPUT y (the sign of y selects the slope, fused_bias_act_kernel.cu:41):
 *   gx = gy * (y > 0 ? 1 : slope) * scale;  gbias[c] = sum_p gx[c][p];  gnoise_weight[0] = sum_{c,p} gx[c][p] * noise[p]
 * gbias / gnoise_weight may be NULL (not needed); they are overwritten.  DETERMINISTIC since round 4: every workgroup stores its partial
 * sums in `partials` (ag_noise_bias_act_partial_floats(C, HW) floats of caller-owned scratch, required when either sum is wanted) and a
 * one-workgroup finish adds them in a fixed order -- bit-identical from run to run given the same inputs (float atomics before).
 * `partials` holds (sum, sum * noise) per workgroup, then one float per workgroup: the largest |gx| it wrote (used by the grouped layer
 * calls, include/ag_layers.h: the fp16-split convolutions that consume gx need its largest magnitude and do not sweep it again).
 */
size_t ag_noise_bias_act_partial_floats(int32_t C, int32_t HW);
int ag_noise_bias_act_backward(float* gx, const float* gy, const float* y, const float* noise, float* gbias, float* gnoise_weight,
                               float* partials, int32_t C, int32_t HW, float slope, float scale, void* stream);

/*
 * Weight modulation + demodulation of ModulatedConv2d's fused branch (network/styleunet/dual_styleunet.py:254-259):
 *   w'[co][ci][k] = (scale * W[co][ci][k]) * style[ci];  dcoef[co] = rsqrt(sum_{ci,k} w'^2 + 1e-8)  (1 when !demodulate)
 *   out = w' * dcoef[co],  laid out [Co][Ci][K2], or [Ci][Co][K2] when `transposed` (what conv_transpose2d takes, :268-272).
 * W [Co][Ci][K2], style [Ci]; dcoef [Co] may be NULL when not needed for a backward.
 */
int ag_modulate_weight_forward(float* out, float* dcoef, const float* W, const float* style, float scale, int32_t demodulate,
                               int32_t Co, int32_t Ci, int32_t K2, int32_t transposed, void* stream);

/* Backward: g = dL/dout (same layout as out).  dW [Co][Ci][K2] and dstyle [Ci] are overwritten.  Deterministic since round 4: the per-(co, ci)
 * tap sums go to `partials` (ag_modulate_weight_partial_floats(Co, Ci) = Co * Ci floats of caller-owned scratch) and are added over co in
 * a fixed order. */
size_t ag_modulate_weight_partial_floats(int32_t Co, int32_t Ci);
int ag_modulate_weight_backward(float* dW, float* dstyle, float* partials, const float* g, const float* W, const float* style,
                                const float* dcoef, float scale, int32_t demodulate, int32_t Co, int32_t Ci, int32_t K2, int32_t transposed,
                                void* stream);

/*
 * 2x2 block transform = HaarTransform / InverseHaarTransform of the wavelet skip path (dual_styleunet.py:374-425: four
 * upfirdn2d calls with 2x2 kernels, down = 2 resp. up = 2, plus a concatenation resp. three additions) in one pass.
 *   merge == 0 (split): in [C][2h][2w] -> out [4][C][h][w],  out[b][c][i][j] = sum_p matrix16[4b + p] * in[c][2i + p/2][2j + p%2]
 *   merge != 0        : in [4][C][h][w] -> out [C][2h][2w],  out[c][2i + p/2][2j + p%2] = sum_b matrix16[4p + b] * in[b][c][i][j]
 * matrix16 is a HOST pointer to the 16 coefficients.  The adjoint of a split with M is a merge with M^T and vice versa.
 */
int ag_block2x2_transform(float* out, const float* in, const float* matrix16, int32_t merge, int32_t C, int32_t h, int32_t w,
                          void* stream);
/* Round 3: ToRGB's wavelet-domain skip path (dual_styleunet.py:607-633: InverseHaarTransform :406-425 -> Upsample :32-50 -> HaarTransform
 * :387-403, added to the layer's output) as one linear pass.  skip [4C, h, w] (sub-band-major channels, as torch.cat builds them) ->
 * out [4C, 2h, 2w]; taps: HOST pointer to the 48 coefficients of the two 1-D factors of the composed map, wy[u'][p][u][a] then
 * wx[u'][p][u][b] (2 x 2 x 2 x 3 each: output sub-band, output parity, input sub-band, site offset + 1; sub-band index of a channel group =
 * uy + 2 ux; derived from the Haar matrices and the FIR kernel by styleunet_ops.skip_chain_taps_1d; passed to the kernel by value);
 * accumulate != 0 adds into out.  backward: the adjoint, gskip [4C, h, w] from gout [4C, 2h, 2w]. */
int ag_skip_chain_forward(float* out, const float* skip, const float* taps, int32_t C, int32_t h, int32_t w, int32_t accumulate, void* stream);
int ag_skip_chain_backward(float* gskip, const float* gout, const float* taps, int32_t C, int32_t h, int32_t w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AG_STYLEUNET_H */
