/*
 * ag_layers.h — layer-level entry points of the StyleUNet (libag_hip.so): ONE native call per ConvLayer / StyledConv, forward and
 * backward, each the fixed sequence of the per-kernel entry points of ag_styleunet.h / ag_conv.h.
 *
 * Replaces, per call, the chains the reference builds out of separate modules:
 *   ConvLayer   = [Blur] + EqualConv2d + FusedLeakyReLU                 network/styleunet/dual_styleunet.py:326-371
 *   StyledConv  = ModulatedConv2d (modulate, demodulate, [transposed] convolution, [Blur]) + NoiseInjection + FusedLeakyReLU
 *                                                                        dual_styleunet.py:225-313,570-604
 * Why: the training iteration is bound by the host (1 300 native calls of 13-32 us per step, profiles/r03_host_vs_gpu.txt); the
 * kernels, their order and therefore the results are exactly those of the per-kernel calls (tests/test_styleunet_net.py compares
 * the two paths bit for bit).  Device pointers, contiguous fp32, batch 1; 0 on success (codes in ag_raster.h).  All scratch is the
 * caller's: `scratch` holds the intermediates of the call (ag_*_scratch_floats), `workspace` is the convolution workspace of
 * ag_conv_workspace_bytes for the layer's convolution.
 */
#ifndef AG_LAYERS_H
#define AG_LAYERS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct AgLayerArgs {
    int32_t Cin, Cout, H, W;       /* input channels / output channels / input spatial size */
    int32_t k;                     /* square kernel size */
    int32_t resample;              /* ConvLayer: 1 = Blur pad (2,2) + stride-2 convolution; StyledConv: 1 = conv_transpose2d stride 2 + Blur pad (1,1) */
    int32_t modulated;             /* 0 ConvLayer (EqualConv2d: weight * scale inside the re-pack), 1 StyledConv */
    int32_t reserved;
    float scale;                   /* EqualConv2d / ModulatedConv2d scale 1 / sqrt(Cin k^2) */
    float slope, act_scale;        /* leaky ReLU slope (0.2) and gain (sqrt 2) */
    float reserved_f;
    const float* x;                /* [Cin, H, W] */
    const float* weight;           /* [Cout, Cin, k, k] */
    const float* style;            /* StyledConv: [Cin] */
    const float* noise;            /* StyledConv: [OH * OW] or NULL */
    const float* noise_weight;     /* StyledConv: [1] or NULL */
    const float* act_bias;         /* [Cout] or NULL */
    const float* k_blur;           /* [4, 4] FIR taps of the layer's Blur (resample only); backward: the FLIPPED taps */
    float* w_mod;                  /* StyledConv: modulated weight [Cout, Cin, k, k] ([Cin, Cout, k, k] when resample), written forward, read backward */
    float* demod;                  /* StyledConv: demodulation coefficients [Cout], written forward, read backward */
    float* x_blur;                 /* down-sampling ConvLayer: the blurred input [Cin, H + 1, W + 1], written forward, read by the weight gradient */
    float* out;                    /* forward: [Cout, OH, OW]; backward: the saved forward output (the activation's backward reads its sign) */
    float* scratch;                /* ag_layer_scratch_floats(args, backward) floats */
    void* workspace;               /* ag_conv_workspace_bytes of the layer's convolution */
    size_t workspace_bytes;
    /* backward only */
    const float* g_out;            /* [Cout, OH, OW] */
    float* g_x;                    /* [Cin, H, W] or NULL */
    float* g_weight;               /* [Cout, Cin, k, k] or NULL (StyledConv: also NULL when neither weight nor style wants a gradient) */
    float* g_style;                /* StyledConv: [Cin] (required with g_weight) */
    float* g_bias_noise;           /* [Cout] bias sums, followed by [1] noise-strength sum when `noise_weight` is set; NULL: no parameter gradient */
    int32_t want_bias, want_noise_weight;
} AgLayerArgs;

/* sizeof(AgLayerArgs) as the library was compiled: a binding checks its own struct layout against it. */
size_t ag_layer_args_bytes(void);
/* Output spatial size of the layer. */
int ag_layer_output_size(const AgLayerArgs* a, int32_t* OH, int32_t* OW);
/* Floats of `scratch` the forward (backward = 0) or backward (1) call needs. */
size_t ag_layer_scratch_floats(const AgLayerArgs* a, int32_t backward);

int ag_layer_forward(const AgLayerArgs* a, void* stream);
int ag_layer_backward(const AgLayerArgs* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AG_LAYERS_H */
