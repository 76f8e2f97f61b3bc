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
    float* w_mod;                  /* StyledConv: modulated weight [Cout, Cin, k, k] (also when resample: the transposed convolution reads this
                                      layout directly), written forward, read backward */
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

/*
 * Grouped layers (round 4): G instances of one layer shape in ONE call and one launch per kernel kind.
 *
 * The avatar runs three DualStyleUNets with identical layer shapes on the same pose map (network/avatar.py:34-36,93-124), each with two
 * decoders of identical shapes (dual_styleunet.py:869-905): six instances of every decoder layer, three of every encoder layer, that
 * differ only in their parameter tensors.  Activations are stacked [G][C][H][W]; the parameters stay the reference's separate tensors and
 * are passed as tables of G device pointers (no copies, gradients are written stacked [G][...] so that each instance's slice is a dense
 * tensor of its parameter's shape).  AgLayerArgs above is the G = 1 case and runs through the same code.
 */
#define AG_MAX_GROUPS 16

typedef struct AgGroupedLayerArgs {
    int32_t G;                     /* instances, 1 .. AG_MAX_GROUPS */
    int32_t Cin, Cout, H, W, k, resample, modulated;   /* as AgLayerArgs */
    float scale, slope, act_scale, reserved_f;
    const float* x;                /* [G][Cin][H][W] */
    int64_t x_group_stride;        /* floats between the instances' inputs: Cin*H*W, or 0 = ONE input [Cin][H][W] shared by all instances
                                      (the pose map / its image pyramid; g_x must then be NULL).  Ignored when G = 1 */
    const float* weight[AG_MAX_GROUPS];        /* per instance: [Cout, Cin, k, k] */
    const float* style[AG_MAX_GROUPS];         /* StyledConv: [Cin] */
    const float* noise[AG_MAX_GROUPS];         /* StyledConv: [OH * OW] or NULL */
    const float* noise_weight[AG_MAX_GROUPS];  /* StyledConv: [1] or NULL */
    const float* act_bias[AG_MAX_GROUPS];      /* [Cout] or NULL */
    const float* k_blur;           /* [4, 4] FIR taps (resample only); backward: the FLIPPED taps */
    float* w_mod;                  /* StyledConv: [G][Cout * Cin * k * k] modulated weights (per instance laid out as AgLayerArgs.w_mod) */
    float* demod;                  /* StyledConv: [G][Cout] */
    float* x_blur;                 /* down-sampling ConvLayer: [G][Cin][H + 1][W + 1] ([Cin][H + 1][W + 1] when the input is shared) */
    float* out;                    /* [G][Cout][OH][OW] */
    float* scratch;                /* ag_grouped_layer_scratch_floats(args, backward) floats */
    void* workspace;               /* ag_grouped_layer_workspace_bytes(args) bytes */
    size_t workspace_bytes;
    /* backward only */
    const float* g_out;            /* [G][Cout][OH][OW] */
    float* g_x;                    /* [G][Cin][H][W] or NULL */
    float* g_weight;               /* [G][Cout][Cin][k][k] or NULL */
    float* g_style;                /* StyledConv: [G][Cin] (required with g_weight) */
    float* g_bias_noise;           /* [G][Cout + 1]: the bias sums of an instance, then its noise-strength sum; NULL: no parameter gradient */
    int32_t want_bias, want_noise_weight;
    float* operand_maxima;         /* ag_grouped_layer_maxima_floats() floats or NULL.  AG_CONV_MATH_SPLIT_F16 (include/ag_conv.h) needs the largest
                                      magnitude of every convolution operand: the forward leaves those of its weights and its input here, the
                                      backward of the SAME layer call reads them instead of sweeping both tensors again.  NULL: every call
                                      takes its own.  Untouched in the other arithmetic modes and by 1 x 1 layers. */
    const float* x_maxima;         /* AG_MAX_GROUPS * 256 floats or NULL: the largest magnitudes of `x` as a previous call's out_maxima left them (instance
                                      g at g * 256; a shared input at 0) -- forward and backward then do not sweep x (ignored by the down-sampling
                                      ConvLayer, whose convolution reads the blurred input, and outside the scaled fp16 modes).  The forward also copies
                                      them into operand_maxima's input slot, so a backward call may pass operand_maxima with or without x_maxima.
                                      They MUST be current: a maximum that is too small overflows fp16 (reported through ag_conv_status, AG_ERR_RANGE) */
    float* out_maxima;             /* forward: AG_MAX_GROUPS * 256 floats or NULL: receives the largest magnitudes of `out` (from the kernel that writes
                                      it wherever that kernel can, by a sweep otherwise; untouched outside AG_CONV_MATH_SPLIT_F16) */
    /* Frozen weights (round 5; forward only): inference runs the same weights and styles frame after frame (main_avatar.py:525-776), and their
       modulation, maxima and tile-blocked fp16 image are a seventh of a frame's device time.  `packed_weights`: ag_grouped_layer_packed_bytes()
       bytes the caller keeps between calls; the call leaves the packed convolution weights there.  `weights_cached` = 1: w_mod / demod, the weight
       slot of operand_maxima and packed_weights ARE those a previous call with the same weights, styles and arithmetic mode left -- the call
       launches no modulation, no weight sweep and no pack kernel.  (Nothing checks that they are current: the caller's contract.) */
    void* packed_weights;
    int32_t weights_cached, reserved_i;
} AgGroupedLayerArgs;

size_t ag_grouped_layer_args_bytes(void);
int ag_grouped_layer_output_size(const AgGroupedLayerArgs* a, int32_t* OH, int32_t* OW);
size_t ag_grouped_layer_scratch_floats(const AgGroupedLayerArgs* a, int32_t backward);
size_t ag_grouped_layer_workspace_bytes(const AgGroupedLayerArgs* a);
size_t ag_grouped_layer_maxima_floats(void);
size_t ag_grouped_layer_packed_bytes(const AgGroupedLayerArgs* a);
int ag_grouped_layer_forward(const AgGroupedLayerArgs* a, void* stream);
int ag_grouped_layer_backward(const AgGroupedLayerArgs* a, void* stream);

/*
 * ToRGB (dual_styleunet.py:607-633) for G instances: modulated 1 x 1 convolution without demodulation + bias, plus the wavelet-domain
 * up-sampled skip (InverseHaarTransform -> Upsample -> HaarTransform, one kernel) accumulated into the output.
 * The bias gradient (sum of g_out over the pixels) is left to the caller.
 */
typedef struct AgGroupedToRgbArgs {
    int32_t G, Cin, Cout, H, W;
    float scale;                   /* 1 / sqrt(Cin) */
    const float* x;                /* [G][Cin][H][W] */
    const float* weight[AG_MAX_GROUPS];   /* [Cout, Cin] */
    const float* style[AG_MAX_GROUPS];    /* [Cin] */
    const float* bias[AG_MAX_GROUPS];     /* [Cout] or NULL */
    const float* skip;             /* [G][Cout][H / 2][W / 2] or NULL */
    const float* skip_taps;        /* HOST pointer to the 48 coefficients of ag_skip_chain_forward (with skip / g_skip) */
    float* w_mod;                  /* [G][Cout * Cin]: written forward, read backward */
    float* out;                    /* [G][Cout][H][W] */
    float* scratch;                /* ag_grouped_to_rgb_scratch_floats floats */
    void* workspace;               /* ag_grouped_to_rgb_workspace_bytes bytes */
    size_t workspace_bytes;
    /* backward only */
    const float* g_out;            /* [G][Cout][H][W] */
    float* g_x;                    /* [G][Cin][H][W] or NULL */
    float* g_weight;               /* [G][Cout][Cin] or NULL */
    float* g_style;                /* [G][Cin] (required with g_weight) */
    float* g_skip;                 /* [G][Cout][H / 2][W / 2] or NULL */
    int32_t weights_cached, reserved_i;   /* forward: 1 = w_mod already holds these weights' modulation (as AgGroupedLayerArgs.weights_cached) */
} AgGroupedToRgbArgs;

size_t ag_grouped_to_rgb_args_bytes(void);
size_t ag_grouped_to_rgb_scratch_floats(const AgGroupedToRgbArgs* a, int32_t backward);
size_t ag_grouped_to_rgb_workspace_bytes(const AgGroupedToRgbArgs* a);
int ag_grouped_to_rgb_forward(const AgGroupedToRgbArgs* a, void* stream);
int ag_grouped_to_rgb_backward(const AgGroupedToRgbArgs* a, void* stream);

/*
 * The comb convolution of a decoder stage (dual_styleunet.py:877-879: ``comb_convs[..](cat([out, cond_list[..]], 1))``, a 3 x 3 ConvLayer
 * with bias + leaky ReLU) for M stacked decoder instances of N networks, WITHOUT building the concatenation:
 *     conv(cat(out_m, lev_r), W_r) = conv(out_m, W_r[:, :C1]) + conv(lev_r, W_r[:, C1:])
 * and the second term depends on the network r only -- both branches of a network (and every camera view of the colour network) read the same
 * encoder level through the same weights -- so it is computed once per network (t) and added inside the activation kernel of every
 * member: a quarter of the comb convolutions' FLOPs at two members per network, forward and backward (the gradients of the members of a
 * network are summed BEFORE the level half's input / weight gradient).  Same values up to the association of the sum over the input
 * channels (the two halves are accumulated separately and added in fp32).
 */
typedef struct AgGroupedCombArgs {
    int32_t M, N;                  /* decoder instances, networks */
    int32_t C1, C2, Cout, H, W;    /* channels of the members' input / of the encoder level / of the output; 3 x 3, stride 1, padding 1 */
    int32_t member_begin[AG_MAX_GROUPS + 1];   /* the members of network r are [member_begin[r], member_begin[r + 1]) (consecutive, non-empty) */
    float scale, slope, act_scale, reserved_f;   /* EqualConv2d scale 1 / sqrt((C1 + C2) * 9); leaky ReLU slope and gain */
    const float* x;                /* [M][C1][H][W] */
    const float* lev;              /* [N][C2][H][W] */
    const float* weight[AG_MAX_GROUPS];     /* per NETWORK: the comb weight [Cout][C1 + C2][3][3] */
    const float* act_bias[AG_MAX_GROUPS];   /* per MEMBER: [Cout] (the members of a network pass the same pointer) */
    float* out;                    /* [M][Cout][H][W] */
    float* scratch;                /* ag_grouped_comb_scratch_floats floats */
    void* workspace;               /* ag_grouped_comb_workspace_bytes bytes */
    size_t workspace_bytes;
    /* backward only */
    const float* g_out;            /* [M][Cout][H][W] */
    float* g_x;                    /* [M][C1][H][W] or NULL */
    float* g_lev;                  /* [N][C2][H][W] or NULL */
    float* g_weight_x;             /* [M][Cout][C1][3][3]: per member, w.r.t. W_r[:, :C1] (NULL: no weight gradients) */
    float* g_weight_lev;           /* [N][Cout][C2][3][3]: per network, w.r.t. W_r[:, C1:] (required with g_weight_x) */
    float* g_bias;                 /* [M][Cout] or NULL */
    float* g_weight;               /* [N][Cout][C1 + C2][3][3] or NULL: the weight gradients in the PARAMETERS' own layout (zeroed by the call; the
                                      members of a network accumulate into its tensor).  When given, g_weight_x / g_weight_lev are not used */
    float* operand_maxima;         /* ag_grouped_comb_maxima_floats() floats or NULL: as AgGroupedLayerArgs.operand_maxima (written by the forward,
                                      read by the backward of the same call) */
    const float* x_maxima;         /* as AgGroupedLayerArgs.x_maxima, for the members' input `x` */
    float* out_maxima;             /* as AgGroupedLayerArgs.out_maxima */
    void* packed_x;                /* frozen weights, as AgGroupedLayerArgs.packed_weights: ag_grouped_comb_packed_bytes(a, 0) bytes for W[:, :C1] ... */
    void* packed_lev;              /* ... and ag_grouped_comb_packed_bytes(a, 1) bytes for W[:, C1:] */
    int32_t weights_cached, reserved_i;   /* 1: both images and the two weight slots of operand_maxima are a previous call's */
} AgGroupedCombArgs;

size_t ag_grouped_comb_args_bytes(void);
size_t ag_grouped_comb_scratch_floats(const AgGroupedCombArgs* a, int32_t backward);
size_t ag_grouped_comb_workspace_bytes(const AgGroupedCombArgs* a);
size_t ag_grouped_comb_maxima_floats(void);
size_t ag_grouped_comb_packed_bytes(const AgGroupedCombArgs* a, int32_t level_half);
int ag_grouped_comb_forward(const AgGroupedCombArgs* a, void* stream);
int ag_grouped_comb_backward(const AgGroupedCombArgs* a, void* stream);

/* ag_block2x2_transform (ag_styleunet.h) on G stacked tensors: in [G][C][2h][2w] <-> out [G][4][C][h][w]. */
int ag_grouped_block2x2(float* out, const float* in, const float* matrix16, int32_t merge, int32_t G, int32_t C, int32_t h, int32_t w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AG_LAYERS_H */
