/*
 * ag_lpips.h — C ABI of the two non-convolution pieces of the LPIPS-VGG16 loss (libag_hip.so), batch 1, fp32.
 *
 * SURVEY.md §8(f)-1 (the loss tail of the training step, "next" after the render path): the reference computes
 * LPIPS(net='vgg') on a 512^2 crop every iteration (main_avatar.py:117-124,227-238; network/lpips/lpips.py:84-127).
 * The VGG16 trunk (torchvision 0.15.2 `vgg16().features[0:30]`, network/lpips/pretrained_networks.py:97-134) is thirteen
 * 3x3 convolutions with bias + ReLU -- ag_conv.h + ag_noise_bias_act_* with slope 0 -- and four 2x2 max-pools; on top of it
 * LPIPS unit-normalises the features of both images over channels, squares their difference, weights the channels with a
 * learned non-negative vector (the 1x1 "lin" convolution) and averages over pixels.  Device pointers, contiguous CHW;
 * 0 on success (codes in ag_raster.h).
 */
#ifndef AG_LPIPS_H
#define AG_LPIPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* nn.MaxPool2d(kernel_size=2, stride=2) (floor mode): x [C][H][W] -> y [C][H/2][W/2]; arg [C][H/2][W/2] (uint8, 0..3 =
 * 2*dy + dx of the first maximum in row-major order, as torch resolves ties) is written when non-NULL for the backward. */
int ag_maxpool2x2_forward(float* y, uint8_t* arg, const float* x, int32_t C, int32_t H, int32_t W, void* stream);

/* gx [C][H][W] = scatter of gy [C][H/2][W/2] to the recorded maxima, zero elsewhere (every element of gx is written). */
int ag_maxpool2x2_backward(float* gx, const float* gy, const uint8_t* arg, int32_t C, int32_t H, int32_t W, void* stream);

/*
 * One LPIPS level (lpips.py:93-103 with spatial = False; normalize_tensor = network/lpips/__init__.py:40-42):
 *   n_i[p] = sqrt(sum_c f_i[c][p]^2 + 1e-10);  a = f0 / (n0 + 1e-10), b = f1 / (n1 + 1e-10)
 *   out[0] += (1 / HW) * sum_p sum_c lin[c] * (a[c][p] - b[c][p])^2
 * f0, f1 [C][HW]; lin [C]; out: one float that the call ACCUMULATES into (zero it once, then call per level).
 */
int ag_lpips_level_forward(float* out, const float* f0, const float* f1, const float* lin, int32_t C, int32_t HW, void* stream);

/* Gradient of the above with respect to f0 only (f1 is the ground-truth image's features): gf0 [C][HW] is overwritten with
 * gout[0] * d out / d f0 (gout: device pointer to the upstream scalar gradient). */
int ag_lpips_level_backward(float* gf0, const float* gout, const float* f0, const float* f1, const float* lin, int32_t C, int32_t HW,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AG_LPIPS_H */
