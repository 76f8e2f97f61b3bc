// Grouped ("batched, per-sample weights") execution of the StyleUNet layers (gfx950).
//
// The avatar evaluates three DualStyleUNets of identical layer shapes on the same pose map (network/avatar.py:34-36,93-124) and each
// of them runs two decoders of identical shapes (dual_styleunet.py:869-905): up to G = 6 (more with several camera views of a pose)
// instances of every layer that differ only in their parameter tensors.  A grouped launch runs the G instances of one layer as ONE
// kernel: activations are stacked [G][C][H][W] (a group stride between instances; stride 0 = one input shared by all), parameters stay
// the reference's separate tensors and reach the kernel as a table of G pointers inside the kernel-argument structure (uniform per
// workgroup: a scalar load from the kernarg segment).  G = 1 is the plain single-instance call: the per-kernel ABI of
// include/ag_styleunet.h / ag_conv.h is the G = 1 case of these launchers, there is one implementation of every kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace ag {

constexpr int kMaxGroups = 16;

struct PtrTable {
    const float* p[kMaxGroups];
};

inline PtrTable table_of(const float* one)
{
    PtrTable t{};
    t.p[0] = one;
    return t;
}

inline PtrTable table_of(const float* const* list, int G)
{
    PtrTable t{};
    for (int g = 0; g < G && g < kMaxGroups; g++) t.p[g] = list ? list[g] : nullptr;
    return t;
}

inline bool table_complete(const PtrTable& t, int G)
{
    for (int g = 0; g < G; g++)
        if (!t.p[g]) return false;
    return true;
}

// ---- ag_styleunet_ops.hip --------------------------------------------------------------------------------------------------------
// y [G][C][HW] = lrelu(x + nw_g[0] * noise_g[pix] + bias_g[c]) * scale; a null table entry = no noise / no bias for that instance
int noise_bias_act_forward_g(float* y, const float* x, int G, const PtrTable& noise, const PtrTable& nw, const PtrTable& bias, int C, int HW,
                             float slope, float scale, hipStream_t s);
// the same with a per-instance addend [C][HW] inside the activation: y = lrelu(x + addend_g + bias_g[c]) * scale (null entry: none)
int bias_act_forward_addend_g(float* y, const float* x, int G, const PtrTable& addend, const PtrTable& bias, int C, int HW, float slope, float scale,
                              hipStream_t s);
// out[r] = sum of in[m] over the members m in [begin[r], begin[r + 1]) -- instances of n floats each, R <= kMaxGroups ranges
int sum_member_ranges(float* out, const float* in, const int* begin, int R, long long n, hipStream_t s);
// floats of `partials` the backward needs
size_t noise_bias_act_partial_floats(int G, int C, int HW);
// gbias: instance g writes C sums at gbias + g * gb_stride (null: not wanted); gnw: one sum at gnw + g * gnw_stride (null: not wanted).
// Deterministic: per-workgroup partial sums + a fixed-order finish (no float atomics).
int fir4x4_amax_g(float* out, const float* in, const float* taps, int planes, int in_h, int in_w, int pad, float* amax, int planes_per_group,
                  hipStream_t s);
int blur_act_forward_g(float* y, const float* x, const float* taps, int G, const PtrTable& noise, const PtrTable& nw, const PtrTable& bias,
                       int C, int H, int W, float slope, float scale, hipStream_t s, float* out_amax = nullptr);
int noise_bias_act_backward_g(float* gx, const float* gy, const float* y, int G, const PtrTable& noise, float* gbias, long long gb_stride,
                              float* gnw, long long gnw_stride, float* partials, int C, int HW, float slope, float scale, hipStream_t s, float* amax = nullptr);
// out [G][Co][Ci][K2] (or [G][Ci][Co][K2] transposed), dcoef [G][Co] (may be null)
int modulate_weight_forward_g(float* out, float* dcoef, int G, const PtrTable& W, const PtrTable& style, float scale, int demod, int Co, int Ci,
                              int K2, int transposed, hipStream_t s, float* rowmax = nullptr);
size_t modulate_weight_partial_floats(int G, int Co, int Ci);
// dW [G][Co][Ci][K2], dstyle [G][Ci], g = dL/dout stacked like out; deterministic (partials [G][Co][Ci] + fixed-order finish)
int modulate_weight_backward_g(float* dW, float* dstyle, float* partials, const float* g, int G, const PtrTable& W, const PtrTable& style,
                               const float* dcoef, float scale, int demod, int Co, int Ci, int K2, int transposed, hipStream_t s);
int skip_chain_forward_g(float* out, const float* skip, const float* taps_host, int G, int C, int h, int w, int accumulate, hipStream_t s);
int skip_chain_backward_g(float* gskip, const float* gout, const float* taps_host, int G, int C, int h, int w, hipStream_t s);
int block2x2_transform_g(float* out, const float* in, const float* matrix16_host, int merge, int G, int C, int h, int w, hipStream_t s);

// ---- ag_conv.hip -----------------------------------------------------------------------------------------------------------------
// Strides in floats between the instances' tensors; x_gs = 0 shares one input among all instances.  Weights (and the forward's bias) come
// from pointer tables; the G weight gradients are written stacked at dw_gs floats.  out_scale is a single-instance option.
}  // namespace ag
struct AgConvDesc;
namespace ag {
// Activation applied inside a forward convolution's epilogue (or its split-K finish): y = lrelu((acc + nw_g[0] * noise_g[pix]) + bias_g[m], slope)
// * scale (kind 1, the StyledConv / ConvLayer tail: dual_styleunet.py:598-604,367-369; a null noise entry = no noise) or
// y = lrelu((acc + addend_g[m][pix]) + bias_g[m], slope) * scale (kind 2: a comb convolution's level half).  The bias is the call's bias table.
// Same expression, same bits as noise_bias_act_forward_kernel on the stored pre-activation tensor.
struct ConvAct {
    int kind;
    float slope, scale;
    PtrTable noise;          // kind 1: noise maps; kind 2: addends
    PtrTable nw;             // kind 1: noise weights
    float* out_amax = nullptr;   // [G][kAmaxParts] zeroed slots or null: the largest magnitude of the activated output of every instance is left there
                                 // (unsigned atomic maxima of the float bits) -- the operand maximum of the NEXT convolution's fp16 split form
};

// fp16 split form of the convolutions (AG_CONV_MATH_SPLIT_F16): every operand tensor's largest magnitude as kAmaxParts partial maxima per
// instance, finished by the consuming kernel.  A call computes what it is not given (ConvOpts::amax_*); a caller that runs several calls
// on the same tensors (a layer's backward: dL/dx and dL/dw share dy) computes them once with conv_absmax.
constexpr int kAmaxParts = 256;
struct AmaxTensor {                 // instances: `table` entries, or ptr + g * gs, or ONE shared instance (no table, gs == 0); all null: skipped
    const float* ptr;
    const PtrTable* table;
    long long gs, len, stride;      // an instance = `rows` runs of `len` floats, `stride` floats apart
    int rows;
    int inst;                       // instances of this tensor when it differs from the call's G (0: G)
};
// tensor i's instance g -> out[(i * kMaxGroups + g) * kAmaxParts ...]; one launch
// zero / zero_inst: additionally zeroes zero[0 .. zero_inst)[kAmaxParts] in the same launch (slots a producer kernel then raises, ConvAct::out_amax)
int conv_absmax(const AmaxTensor* t, int n, int G, float* out, hipStream_t s, float* zero = nullptr, int zero_inst = 0);
size_t conv_absmax_floats(int tensors);
bool conv_math_needs_absmax();

// Options of a grouped convolution call.
struct ConvOpts {
    const ConvAct* act = nullptr;   // forward, plain gathers (not the transposed convolution) only
    bool wt_oihw = false;     // the weights (and weight gradients) of a TRANSPOSED convolution are laid out [Cout][Cin][k][k] like a convolution's
                              // instead of conv_transpose2d's [Cin][Cout][k][k] (ignored for AG_CONV)
    int w_cin_total = 0;      // forward / input gradient of an AG_CONV on a channel SLICE of a wider weight tensor [Cout][w_cin_total][k][k]: the
                              // weight pointers point at the slice's first channel, rows are w_cin_total * k * k floats apart.  0: exactly d->Cin
                              // channels.  (The weight gradient does not read the weights: it is always written [Cout][d->Cin][k][k].)
    // fp16 split form: partial maxima ([kMaxGroups][kAmaxParts], conv_absmax layout of ONE tensor) the caller already has for the weights,
    // the input x and the output gradient dy -- each call uses the two that are its operands and computes the ones left null
    const float* amax_w = nullptr;
    const float* amax_x = nullptr;
    const float* amax_dy = nullptr;
    // weight gradient written into a wider tensor (the comb convolutions: a channel slice of the parameter's own gradient): instance g goes to
    // dw_table->p[g] with rows dw_row_stride floats apart; CONSECUTIVE instances with the same entry are summed into it (in instance order, by the
    // weight gradient's slice reduction); the slice is overwritten
    const PtrTable* dw_table = nullptr;
    long long dw_row_stride = 0;
    // Frozen weights (round 5, inference): `packed` = conv_packed_bytes_g(d, G) bytes the CALLER keeps between calls; the forward / input-gradient
    // call packs the weights into it instead of its workspace, or -- packed_valid -- finds them there and launches no pack kernel (the weights,
    // their scale and the arithmetic mode must be those of the call that packed them; with the scaled fp16 forms amax_w must be given too)
    float* packed = nullptr;
    bool packed_valid = false;
};
size_t conv_workspace_bytes_g(const AgConvDesc* d, int G);
size_t conv_packed_bytes_g(const AgConvDesc* d, int G);
int conv_forward_g(const AgConvDesc* d, int G, const float* x, long long x_gs, const PtrTable& w, const float* out_scale, const PtrTable& bias,
                   float* y, long long y_gs, void* workspace, size_t workspace_bytes, hipStream_t s, const ConvOpts& o = ConvOpts());
int conv_backward_input_g(const AgConvDesc* d, int G, const float* dy, long long dy_gs, const PtrTable& w, float* dx, long long dx_gs,
                          void* workspace, size_t workspace_bytes, hipStream_t s, const ConvOpts& o = ConvOpts());
// dw: the G gradients at dw_gs floats (stacked when dw_gs = the size of one weight); overwritten
int conv_backward_weight_g(const AgConvDesc* d, int G, const float* x, long long x_gs, const float* dy, long long dy_gs, float* dw, long long dw_gs,
                           void* workspace, size_t workspace_bytes, hipStream_t s, const ConvOpts& o = ConvOpts());

}  // namespace ag
