// Layer-level entry points (include/ag_layers.h): one native call per ConvLayer / StyledConv / ToRGB, forward and backward, for G
// instances of the layer at once (ag_groups.h; G = 1 is the single-layer call).  Pure composition of the grouped kernel launchers -- same
// kernels, same order as the per-kernel calls -- so that the host pays one Python -> C transition per layer (and per GROUP of layers)
// instead of three to five per layer.
#include "ag_common.h"
#include "ag_groups.h"
#include "../../include/ag_conv.h"
#include "../../include/ag_layers.h"
#include "../../include/ag_styleunet.h"

#include <cstdlib>
#include <cstring>

using namespace ag;

namespace {

struct Geo {
    int OH, OW;          // layer output
    int CH, CW;          // convolution output (before the Blur of an up-sampling StyledConv)
    int BH, BW;          // blurred input of a down-sampling ConvLayer
    AgConvDesc d;
};

bool geometry(const AgGroupedLayerArgs* a, Geo& g)
{
    if (!a || a->G < 1 || a->G > AG_MAX_GROUPS || a->Cin <= 0 || a->Cout <= 0 || a->H <= 0 || a->W <= 0 || a->k <= 0) return false;
    g.BH = a->H; g.BW = a->W;
    g.d = AgConvDesc{};
    g.d.Cin = a->Cin; g.d.Cout = a->Cout; g.d.k = a->k;
    if (!a->modulated) {
        g.d.kind = AG_CONV;
        g.d.weight_scale = a->scale;
        if (a->resample) { g.BH = a->H + 1; g.BW = a->W + 1; g.d.H = g.BH; g.d.W = g.BW; g.d.stride = 2; g.d.padding = 0; }
        else { g.d.H = a->H; g.d.W = a->W; g.d.stride = 1; g.d.padding = a->k / 2; }
    } else {
        g.d.weight_scale = 1.0f;
        g.d.H = a->H; g.d.W = a->W;
        if (a->resample) { g.d.kind = AG_CONV_TRANSPOSE; g.d.stride = 2; g.d.padding = 0; }
        else { g.d.kind = AG_CONV; g.d.stride = 1; g.d.padding = a->k / 2; }
    }
    int32_t oh = 0, ow = 0;
    if (ag_conv_output_size(&g.d, &oh, &ow) != AG_OK) return false;
    g.CH = oh; g.CW = ow;
    if (a->modulated && a->resample) { g.OH = oh - 1; g.OW = ow - 1; }      // Blur pad (1,1) with 4 taps: n + 2 - 4 + 1
    else { g.OH = oh; g.OW = ow; }
    return g.OH > 0 && g.OW > 0;
}

size_t pad64(size_t n) { return (n + 63) / 64 * 64; }

const ConvOpts kOihw = [] { ConvOpts o; o.wt_oihw = true; return o; }();

// The activation pass (noise + bias + leaky ReLU) inside the convolution's epilogue wherever no Blur sits between the two (round 4: the
// pre-activation tensor is neither written nor re-read).  AG_FUSED_ACT=0: the separate pass (same bits; same-box A/B, profiles/).
bool fused_act()
{
    static const bool on = [] { const char* e = getenv("AG_FUSED_ACT"); return !(e && e[0] == '0'); }();
    return on;
}

// float offsets of the regions inside `scratch`
struct Scratch {
    size_t pre, aux, g_blur, g_wm, nba_part, mod_part, amax, rowmax, total;
};

Scratch scratch_layout(const AgGroupedLayerArgs* a, const Geo& g, bool backward)
{
    const size_t G = a->G;
    Scratch s{};
    size_t o = 0;
    s.pre = o;  o += pad64(G * a->Cout * g.OH * g.OW);                                           // activation input / its gradient
    s.aux = o;
    if (a->modulated && a->resample) o += pad64(G * a->Cout * g.CH * g.CW);                        // transposed-convolution output / its gradient
    s.g_blur = o;
    if (!a->modulated && a->resample && backward) o += pad64(G * a->Cin * g.BH * g.BW);            // gradient of the blurred input
    s.g_wm = o;
    if (backward && a->modulated) o += pad64(G * (size_t)a->Cout * a->Cin * a->k * a->k);          // gradient of the modulated weight
    s.nba_part = o;
    if (backward) o += pad64(noise_bias_act_partial_floats(a->G, a->Cout, g.OH * g.OW));
    s.mod_part = o;
    if (backward && a->modulated) o += pad64(modulate_weight_partial_floats(a->G, a->Cout, a->Cin));
    s.amax = o;
    if (backward) o += pad64(conv_absmax_floats(3));                                               // fp16 split form: maxima of dy, w, x
    s.rowmax = o;
    if (!backward && a->modulated) o += pad64(G * a->Cout);                                        // the modulated weight's row maxima
    s.total = o + 64;
    return s;
}

void to_grouped(const AgLayerArgs* a, AgGroupedLayerArgs& g)
{
    memset(&g, 0, sizeof(g));
    g.G = 1;
    g.Cin = a->Cin; g.Cout = a->Cout; g.H = a->H; g.W = a->W; g.k = a->k; g.resample = a->resample; g.modulated = a->modulated;
    g.scale = a->scale; g.slope = a->slope; g.act_scale = a->act_scale;
    g.x = a->x; g.x_group_stride = 0;
    g.weight[0] = a->weight; g.style[0] = a->style; g.noise[0] = a->noise; g.noise_weight[0] = a->noise_weight; g.act_bias[0] = a->act_bias;
    g.k_blur = a->k_blur; g.w_mod = a->w_mod; g.demod = a->demod; g.x_blur = a->x_blur; g.out = a->out;
    g.scratch = a->scratch; g.workspace = a->workspace; g.workspace_bytes = a->workspace_bytes;
    g.g_out = a->g_out; g.g_x = a->g_x; g.g_weight = a->g_weight; g.g_style = a->g_style; g.g_bias_noise = a->g_bias_noise;
    g.want_bias = a->want_bias; g.want_noise_weight = a->want_noise_weight;
}

}  // namespace

extern "C" {

size_t ag_layer_args_bytes(void) { return sizeof(AgLayerArgs); }
size_t ag_grouped_layer_args_bytes(void) { return sizeof(AgGroupedLayerArgs); }
size_t ag_grouped_to_rgb_args_bytes(void) { return sizeof(AgGroupedToRgbArgs); }

int ag_grouped_layer_output_size(const AgGroupedLayerArgs* a, int32_t* OH, int32_t* OW)
{
    Geo g;
    if (!geometry(a, g) || !OH || !OW) { set_error("ag_layer_output_size: bad layer description"); return AG_ERR_INVALID_ARGUMENT; }
    *OH = g.OH; *OW = g.OW;
    return AG_OK;
}

size_t ag_grouped_layer_scratch_floats(const AgGroupedLayerArgs* a, int32_t backward)
{
    Geo g;
    if (!geometry(a, g)) return 0;
    return scratch_layout(a, g, backward != 0).total;
}

size_t ag_grouped_layer_workspace_bytes(const AgGroupedLayerArgs* a)
{
    Geo g;
    if (!geometry(a, g)) return 0;
    return conv_workspace_bytes_g(&g.d, a->G);
}

size_t ag_grouped_layer_maxima_floats(void) { return conv_absmax_floats(2); }

size_t ag_grouped_layer_packed_bytes(const AgGroupedLayerArgs* a)
{
    Geo g;
    if (!geometry(a, g)) return 0;
    return conv_packed_bytes_g(&g.d, a->G);
}

int ag_grouped_layer_forward(const AgGroupedLayerArgs* a, void* stream)
{
    Geo g;
    if (!geometry(a, g) || !a->x || !a->out || !a->scratch) { set_error("ag_layer_forward: bad arguments"); return AG_ERR_INVALID_ARGUMENT; }
    const int G = a->G;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const PtrTable w_t = table_of(a->weight, G), bias_t = table_of(a->act_bias, G);
    if (!table_complete(w_t, G)) { set_error("ag_layer_forward: null weight"); return AG_ERR_INVALID_ARGUMENT; }
    const Scratch L = scratch_layout(a, g, false);
    float* pre = a->scratch + L.pre;
    float* aux = a->scratch + L.aux;
    const long long x_gs = G > 1 ? a->x_group_stride : 0;
    const long long pre_gs = (long long)a->Cout * g.OH * g.OW;
    int rc;
    // the largest magnitudes of this call's output for the NEXT call: from the kernel that writes `out` where it can emit them
    float* const omax = (conv_math_needs_absmax() && a->out_maxima) ? a->out_maxima : nullptr;
    bool omax_zeroed = false;
    // fp16 split form of the MFMA convolutions: the maxima of the forward's operands (weights as convolved, input as convolved) go to
    // a->operand_maxima, where the backward finds them -- its convolutions have the same operands
    const long long w_len = (long long)a->Cout * a->Cin * a->k * a->k;
    const float* w_rowmax = nullptr;      // set by the StyledConv branch: [G][Cout] row maxima of the modulated weights
    auto keep_maxima = [&](const PtrTable& w, const float* x, long long xgs, long long x_len, ConvOpts& o) -> int {
        if (!conv_math_needs_absmax() || a->k < 3 || !a->operand_maxima) return AG_OK;
        AmaxTensor t[2] = { AmaxTensor{ nullptr, &w, 0, w_len, 0, 1 }, AmaxTensor{ x, nullptr, xgs, x_len, 0, 1 } };
        if (w_rowmax) t[0] = AmaxTensor{ w_rowmax, nullptr, a->Cout, a->Cout, 0, 1 };
        if (a->weights_cached) t[0] = AmaxTensor{};          // frozen weights: their slot of operand_maxima is a previous call's
        const bool x_known = a->x_maxima && x == a->x;        // handed over by the call that produced x
        // ... then the x slot of operand_maxima gets the handed maxima themselves ("maxima of the 256 partial maxima": the same largest magnitude, inside
        // the same launch), so that a backward call that is given operand_maxima WITHOUT x_maxima never reads a slot nobody wrote (round-4 advice)
        if (x_known) t[1] = AmaxTensor{ a->x_maxima, nullptr, xgs ? (long long)kAmaxParts : 0, kAmaxParts, 0, 1 };
        const int rc1 = conv_absmax(t, 2, G, a->operand_maxima, s, omax, G);       // (and zeroes the slots of this call's own output maxima)
        if (rc1) return rc1;
        omax_zeroed = true;
        o.amax_w = a->operand_maxima;
        o.amax_x = x_known ? a->x_maxima : a->operand_maxima + (size_t)kMaxGroups * kAmaxParts;
        return AG_OK;
    };
    // frozen weights: the packed image is reused only where its maxima are kept too (1 x 1 layers and calls without operand_maxima take their own)
    const bool packed_frozen = a->weights_cached && a->packed_weights && (!conv_math_needs_absmax() || (a->k >= 3 && a->operand_maxima));
    auto zero_omax = [&]() -> int {       // the slots a producer kernel raises must start at zero: by the maxima launch above, or here
        if (!omax || omax_zeroed) return AG_OK;
        omax_zeroed = true;
        return check_hip(hipMemsetAsync(omax, 0, (size_t)G * kAmaxParts * sizeof(float), s), "memset out_maxima");
    };
    auto sweep_out = [&]() -> int {       // ... by a sweep where no kernel can emit them
        if (!omax) return AG_OK;
        const AmaxTensor t{ a->out, nullptr, pre_gs, pre_gs, 0, 1 };
        return conv_absmax(&t, 1, G, omax, s);
    };
    if (!a->modulated) {
        const float* cx = a->x;
        long long cx_gs = x_gs;
        if (a->resample) {
            if (!a->k_blur || !a->x_blur) { set_error("ag_layer_forward: down-sampling layer without FIR taps / x_blur"); return AG_ERR_INVALID_ARGUMENT; }
            const int n_in = (G > 1 && x_gs == 0) ? 1 : G;                    // a shared input is blurred once
            if ((rc = ag_upfirdn2d(a->x_blur, a->x, a->k_blur, n_in * a->Cin, a->H, a->W, 4, 4, 1, 1, 1, 1, 2, 2, 2, 2, stream))) return rc;
            cx = a->x_blur;
            cx_gs = n_in == 1 ? 0 : (long long)a->Cin * g.BH * g.BW;
        }
        ConvOpts o;
        o.packed = reinterpret_cast<float*>(a->packed_weights); o.packed_valid = packed_frozen;
        if ((rc = keep_maxima(w_t, cx, cx_gs, (long long)a->Cin * (a->resample ? g.BH * g.BW : a->H * a->W), o))) return rc;
        if (fused_act() && !(a->k == 1 && a->Cin <= 4)) {       // (the 3-channel FromRGB convolutions run on the streaming 1 x 1 kernel, which has no such epilogue)
            ConvAct act{ 1, a->slope, a->act_scale, PtrTable{}, PtrTable{} };
            if ((rc = zero_omax())) return rc;
            act.out_amax = omax;
            o.act = &act;
            return conv_forward_g(&g.d, G, cx, cx_gs, w_t, nullptr, bias_t, a->out, pre_gs, a->workspace, a->workspace_bytes, s, o);
        }
        if ((rc = conv_forward_g(&g.d, G, cx, cx_gs, w_t, nullptr, PtrTable{}, pre, pre_gs, a->workspace, a->workspace_bytes, s, o))) return rc;
        if ((rc = noise_bias_act_forward_g(a->out, pre, G, PtrTable{}, PtrTable{}, bias_t, a->Cout, g.OH * g.OW, a->slope, a->act_scale, s))) return rc;
        return sweep_out();
    }
    const PtrTable style_t = table_of(a->style, G);
    if (!table_complete(style_t, G) || !a->w_mod || !a->demod) { set_error("ag_layer_forward: StyledConv needs style, w_mod and demod"); return AG_ERR_INVALID_ARGUMENT; }
    // the modulated weight is kept [Cout][Cin][k][k] for the transposed convolution too (wt_oihw below): the modulation kernels read and
    // write it coalesced both ways (conv_transpose2d's own [Cin][Cout] order made them 9x slower than their bytes, round 4)
    // (rowmax: the modulation kernel leaves every output row's largest magnitude, so the fp16 split form's weight maximum is a sweep of
    // G * Cout numbers instead of the whole modulated weight)
    float* const rowmax = a->scratch + L.rowmax;
    const bool row_maxima = conv_math_needs_absmax() && a->k >= 3 && a->operand_maxima && !a->weights_cached;
    if (!a->weights_cached &&
        (rc = modulate_weight_forward_g(a->w_mod, a->demod, G, w_t, style_t, a->scale, 1, a->Cout, a->Cin, a->k * a->k, 0, s,
                                        row_maxima ? rowmax : nullptr))) return rc;
    // the modulated weights of the instances, stacked
    PtrTable wm_t{};
    const size_t wn = (size_t)a->Cout * a->Cin * a->k * a->k;
    for (int i = 0; i < G; i++) wm_t.p[i] = a->w_mod + i * wn;
    PtrTable noise_t{}, nw_t{};
    for (int i = 0; i < G; i++)
        if (a->noise[i] && a->noise_weight[i]) { noise_t.p[i] = a->noise[i]; nw_t.p[i] = a->noise_weight[i]; }
    if (row_maxima) w_rowmax = rowmax;
    ConvOpts om = kOihw;          // (wt_oihw is ignored by the plain convolution of the non-resampling case)
    om.packed = reinterpret_cast<float*>(a->packed_weights); om.packed_valid = packed_frozen;
    if ((rc = keep_maxima(wm_t, a->x, x_gs, (long long)a->Cin * a->H * a->W, om))) return rc;
    if (a->resample) {
        if (!a->k_blur) { set_error("ag_layer_forward: resampling layer without FIR taps"); return AG_ERR_INVALID_ARGUMENT; }
        if ((rc = conv_forward_g(&g.d, G, a->x, x_gs, wm_t, nullptr, PtrTable{}, aux, (long long)a->Cout * g.CH * g.CW, a->workspace, a->workspace_bytes, s, om))) return rc;
        // Blur + noise + bias + leaky ReLU in one pass (the backward needs the output only, so the blurred pre-activation is never stored)
        if (fused_act() && G * a->Cout <= 65535 && (rc = zero_omax())) return rc;
        if (fused_act() && G * a->Cout <= 65535)
            return blur_act_forward_g(a->out, aux, a->k_blur, G, noise_t, nw_t, bias_t, a->Cout, g.CH, g.CW, a->slope, a->act_scale, s, omax);
        if ((rc = ag_upfirdn2d(pre, aux, a->k_blur, G * a->Cout, g.CH, g.CW, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1, stream))) return rc;
    } else {
        if (fused_act()) {
            ConvAct act{ 1, a->slope, a->act_scale, noise_t, nw_t };
            if ((rc = zero_omax())) return rc;
            act.out_amax = omax;
            om.act = &act;
            return conv_forward_g(&g.d, G, a->x, x_gs, wm_t, nullptr, bias_t, a->out, pre_gs, a->workspace, a->workspace_bytes, s, om);
        }
        if ((rc = conv_forward_g(&g.d, G, a->x, x_gs, wm_t, nullptr, PtrTable{}, pre, pre_gs, a->workspace, a->workspace_bytes, s, om))) return rc;
    }
    if ((rc = noise_bias_act_forward_g(a->out, pre, G, noise_t, nw_t, bias_t, a->Cout, g.OH * g.OW, a->slope, a->act_scale, s))) return rc;
    return sweep_out();
}

int ag_grouped_layer_backward(const AgGroupedLayerArgs* a, void* stream)
{
    Geo g;
    if (!geometry(a, g) || !a->x || !a->out || !a->scratch || !a->g_out) { set_error("ag_layer_backward: bad arguments"); return AG_ERR_INVALID_ARGUMENT; }
    const int G = a->G;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const PtrTable w_t = table_of(a->weight, G);
    if (!table_complete(w_t, G)) { set_error("ag_layer_backward: null weight"); return AG_ERR_INVALID_ARGUMENT; }
    const long long x_gs = G > 1 ? a->x_group_stride : 0;
    if (G > 1 && x_gs == 0 && a->g_x) { set_error("ag_layer_backward: an input shared by the instances has no per-instance gradient"); return AG_ERR_INVALID_ARGUMENT; }
    const Scratch L = scratch_layout(a, g, true);
    float* g_pre = a->scratch + L.pre;
    float* aux = a->scratch + L.aux;
    const long long pre_gs = (long long)a->Cout * g.OH * g.OW;
    PtrTable noise_t{};
    bool all_noise = a->modulated != 0;
    for (int i = 0; i < G; i++) {
        if (a->modulated && a->noise[i] && a->noise_weight[i]) noise_t.p[i] = a->noise[i];
        else all_noise = false;
    }
    float* gb = (a->want_bias && a->g_bias_noise) ? a->g_bias_noise : nullptr;
    float* gnw = (all_noise && a->want_noise_weight && a->g_bias_noise) ? a->g_bias_noise + a->Cout : nullptr;
    int rc;
    // fp16 split form of the MFMA convolutions: the activation backward hands the maximum of the gradient it writes to its consumers
    const bool want_maxima = conv_math_needs_absmax() && a->k >= 3;
    float* const am = a->scratch + L.amax;
    if ((rc = noise_bias_act_backward_g(g_pre, a->g_out, a->out, G, noise_t, gb, a->Cout + 1, gnw, a->Cout + 1, a->scratch + L.nba_part, a->Cout,
                                        g.OH * g.OW, a->slope, a->act_scale, s, want_maxima ? am : nullptr))) return rc;
    // fp16 split form of the MFMA convolutions: dL/dx and dL/dw share dy, so the three operand maxima are taken once, in one launch
    const long long w_len = (long long)a->Cout * a->Cin * a->k * a->k;
    bool dy_maxima_ready = true;       // `am` holds the maxima of g_pre (noise_bias_act_backward_g above); the Blur adjoint below replaces them by its output's
    auto shared_maxima = [&](const float* dy, long long dy_gs, const PtrTable& w, const float* x, long long xgs, long long x_len, ConvOpts& o) -> int {
        if (!want_maxima) return AG_OK;
        AmaxTensor t[3] = { AmaxTensor{ dy, nullptr, dy_gs, dy_gs, 0, 1 }, AmaxTensor{ nullptr, &w, 0, w_len, 0, 1 }, AmaxTensor{ x, nullptr, xgs, x_len, 0, 1 } };
        if (dy_maxima_ready) t[0] = AmaxTensor{};
        const bool x_known = a->x_maxima && x == a->x;                  // handed over by the call that produced x
        if (!a->g_x || a->operand_maxima) t[1] = AmaxTensor{};          // kept by the forward
        if (!a->g_weight || a->operand_maxima || x_known) t[2] = AmaxTensor{};
        const int rc1 = conv_absmax(t, 3, G, am, s);
        if (rc1) return rc1;
        o.amax_dy = am;
        if (a->g_x) o.amax_w = a->operand_maxima ? a->operand_maxima : am + (size_t)kMaxGroups * kAmaxParts;
        if (a->g_weight) o.amax_x = x_known ? a->x_maxima : a->operand_maxima ? a->operand_maxima + (size_t)kMaxGroups * kAmaxParts : am + 2 * (size_t)kMaxGroups * kAmaxParts;
        return AG_OK;
    };
    if (!a->modulated) {
        if (a->resample) {
            if (!a->x_blur || !a->k_blur) { set_error("ag_layer_backward: down-sampling layer without x_blur / FIR taps"); return AG_ERR_INVALID_ARGUMENT; }
            const int n_in = (G > 1 && x_gs == 0) ? 1 : G;
            const long long xb_gs = n_in == 1 ? 0 : (long long)a->Cin * g.BH * g.BW;
            ConvOpts o;
            if ((rc = shared_maxima(g_pre, pre_gs, w_t, a->x_blur, xb_gs, (long long)a->Cin * g.BH * g.BW, o))) return rc;
            if (a->g_x) {     // gradient w.r.t. the blurred input, then the FIR's adjoint (flipped taps, pads (1,1): [H + 1] -> [H])
                float* g_blur = a->scratch + L.g_blur;
                if ((rc = conv_backward_input_g(&g.d, G, g_pre, pre_gs, w_t, g_blur, (long long)a->Cin * g.BH * g.BW, a->workspace, a->workspace_bytes, s, o))) return rc;
                if ((rc = ag_upfirdn2d(a->g_x, g_blur, a->k_blur, G * a->Cin, g.BH, g.BW, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1, stream))) return rc;
            }
            if (a->g_weight && (rc = conv_backward_weight_g(&g.d, G, a->x_blur, xb_gs, g_pre, pre_gs, a->g_weight, (long long)a->Cout * a->Cin * a->k * a->k,
                                                            a->workspace, a->workspace_bytes, s, o))) return rc;
            return AG_OK;
        }
        ConvOpts o;
        if ((rc = shared_maxima(g_pre, pre_gs, w_t, a->x, x_gs, (long long)a->Cin * a->H * a->W, o))) return rc;
        if (a->g_x && (rc = conv_backward_input_g(&g.d, G, g_pre, pre_gs, w_t, a->g_x, (long long)a->Cin * a->H * a->W, a->workspace, a->workspace_bytes, s, o))) return rc;
        if (a->g_weight && (rc = conv_backward_weight_g(&g.d, G, a->x, x_gs, g_pre, pre_gs, a->g_weight, (long long)a->Cout * a->Cin * a->k * a->k, a->workspace,
                                                        a->workspace_bytes, s, o))) return rc;
        return AG_OK;
    }
    const PtrTable style_t = table_of(a->style, G);
    if (!table_complete(style_t, G) || !a->w_mod || !a->demod) { set_error("ag_layer_backward: StyledConv needs style, w_mod and demod"); return AG_ERR_INVALID_ARGUMENT; }
    PtrTable wm_t{};
    const size_t wn = (size_t)a->Cout * a->Cin * a->k * a->k;
    for (int i = 0; i < G; i++) wm_t.p[i] = a->w_mod + i * wn;
    const float* g_conv = g_pre;
    long long gconv_gs = pre_gs;
    if (a->resample) {
        // adjoint of Blur pad (1,1): pads (2,2) with the flipped taps, [2H] -> [2H + 1]; fp16 split form: the kernel leaves the largest
        // magnitude of what it writes in `am` (atomic maxima into zeroed slots) -- the convolutions below do not sweep `aux` for it
        if (want_maxima && G * a->Cout <= 65535) {
            if ((rc = check_hip(hipMemsetAsync(am, 0, (size_t)G * kAmaxParts * sizeof(float), s), "memset maxima"))) return rc;
            if ((rc = fir4x4_amax_g(aux, g_pre, a->k_blur, G * a->Cout, g.OH, g.OW, 2, am, a->Cout, s))) return rc;
        } else {
            if ((rc = ag_upfirdn2d(aux, g_pre, a->k_blur, G * a->Cout, g.OH, g.OW, 4, 4, 1, 1, 1, 1, 2, 2, 2, 2, stream))) return rc;
            dy_maxima_ready = false;
        }
        g_conv = aux;
        gconv_gs = (long long)a->Cout * g.CH * g.CW;
    }
    ConvOpts o = kOihw;
    if ((rc = shared_maxima(g_conv, gconv_gs, wm_t, a->x, x_gs, (long long)a->Cin * a->H * a->W, o))) return rc;
    if (a->g_x && (rc = conv_backward_input_g(&g.d, G, g_conv, gconv_gs, wm_t, a->g_x, (long long)a->Cin * a->H * a->W, a->workspace, a->workspace_bytes, s, o))) return rc;
    if (a->g_weight) {
        if (!a->g_style) { set_error("ag_layer_backward: g_weight without g_style"); return AG_ERR_INVALID_ARGUMENT; }
        float* g_wm = a->scratch + L.g_wm;
        if ((rc = conv_backward_weight_g(&g.d, G, a->x, x_gs, g_conv, gconv_gs, g_wm, (long long)wn, a->workspace, a->workspace_bytes, s, o))) return rc;
        if ((rc = modulate_weight_backward_g(a->g_weight, a->g_style, a->scratch + L.mod_part, g_wm, G, w_t, style_t, a->demod, a->scale, 1, a->Cout, a->Cin,
                                             a->k * a->k, 0, s))) return rc;
    }
    return AG_OK;
}

// ---- single-layer calls = G = 1 ---------------------------------------------------------------------------------------------------
int ag_layer_output_size(const AgLayerArgs* a, int32_t* OH, int32_t* OW)
{
    if (!a) { set_error("ag_layer_output_size: null"); return AG_ERR_INVALID_ARGUMENT; }
    AgGroupedLayerArgs g;
    to_grouped(a, g);
    return ag_grouped_layer_output_size(&g, OH, OW);
}

size_t ag_layer_scratch_floats(const AgLayerArgs* a, int32_t backward)
{
    if (!a) return 0;
    AgGroupedLayerArgs g;
    to_grouped(a, g);
    return ag_grouped_layer_scratch_floats(&g, backward);
}

int ag_layer_forward(const AgLayerArgs* a, void* stream)
{
    if (!a) { set_error("ag_layer_forward: null"); return AG_ERR_INVALID_ARGUMENT; }
    AgGroupedLayerArgs g;
    to_grouped(a, g);
    return ag_grouped_layer_forward(&g, stream);
}

int ag_layer_backward(const AgLayerArgs* a, void* stream)
{
    if (!a) { set_error("ag_layer_backward: null"); return AG_ERR_INVALID_ARGUMENT; }
    AgGroupedLayerArgs g;
    to_grouped(a, g);
    return ag_grouped_layer_backward(&g, stream);
}

// ---- ToRGB (dual_styleunet.py:607-633) for G instances ------------------------------------------------------------------------------
// out = conv1x1(x, (scale * W) * style) + bias [+ HaarTransform(Upsample(InverseHaarTransform(skip)))]
static bool rgb_ok(const AgGroupedToRgbArgs* a)
{
    return a && a->G >= 1 && a->G <= AG_MAX_GROUPS && a->Cin > 0 && a->Cout > 0 && a->H > 0 && a->W > 0;
}

static AgConvDesc rgb_desc(const AgGroupedToRgbArgs* a)
{
    AgConvDesc d{};
    d.kind = AG_CONV; d.Cin = a->Cin; d.Cout = a->Cout; d.H = a->H; d.W = a->W; d.k = 1; d.stride = 1; d.padding = 0; d.weight_scale = 1.0f;
    return d;
}

size_t ag_grouped_to_rgb_workspace_bytes(const AgGroupedToRgbArgs* a)
{
    if (!rgb_ok(a)) return 0;
    const AgConvDesc d = rgb_desc(a);
    return conv_workspace_bytes_g(&d, a->G);
}

size_t ag_grouped_to_rgb_scratch_floats(const AgGroupedToRgbArgs* a, int32_t backward)
{
    if (!rgb_ok(a)) return 0;
    if (!backward) return 64;
    return pad64((size_t)a->G * a->Cout * a->Cin) + pad64(modulate_weight_partial_floats(a->G, a->Cout, a->Cin)) + 64;
}

int ag_grouped_to_rgb_forward(const AgGroupedToRgbArgs* a, void* stream)
{
    if (!rgb_ok(a) || !a->x || !a->out || !a->w_mod) { set_error("ag_to_rgb_forward: bad arguments"); return AG_ERR_INVALID_ARGUMENT; }
    const int G = a->G;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const PtrTable w_t = table_of(a->weight, G), style_t = table_of(a->style, G), bias_t = table_of(a->bias, G);
    int rc;
    if (!a->weights_cached && (rc = modulate_weight_forward_g(a->w_mod, nullptr, G, w_t, style_t, a->scale, 0, a->Cout, a->Cin, 1, 0, s))) return rc;
    PtrTable wm_t{};
    for (int i = 0; i < G; i++) wm_t.p[i] = a->w_mod + (size_t)i * a->Cout * a->Cin;
    const AgConvDesc d = rgb_desc(a);
    const long long hw = (long long)a->H * a->W;
    if ((rc = conv_forward_g(&d, G, a->x, a->Cin * hw, wm_t, nullptr, bias_t, a->out, a->Cout * hw, a->workspace, a->workspace_bytes, s))) return rc;
    if (a->skip) {
        if (!a->skip_taps || (a->Cout & 3) || (a->H & 1) || (a->W & 1)) { set_error("ag_to_rgb_forward: skip path needs taps, 4C channels, even size"); return AG_ERR_INVALID_ARGUMENT; }
        return skip_chain_forward_g(a->out, a->skip, a->skip_taps, G, a->Cout / 4, a->H / 2, a->W / 2, 1, s);
    }
    return AG_OK;
}

int ag_grouped_to_rgb_backward(const AgGroupedToRgbArgs* a, void* stream)
{
    if (!rgb_ok(a) || !a->x || !a->g_out || !a->w_mod || !a->scratch) { set_error("ag_to_rgb_backward: bad arguments"); return AG_ERR_INVALID_ARGUMENT; }
    const int G = a->G;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const PtrTable w_t = table_of(a->weight, G), style_t = table_of(a->style, G);
    PtrTable wm_t{};
    for (int i = 0; i < G; i++) wm_t.p[i] = a->w_mod + (size_t)i * a->Cout * a->Cin;
    const AgConvDesc d = rgb_desc(a);
    const long long hw = (long long)a->H * a->W;
    int rc;
    if (a->g_skip) {
        if (!a->skip_taps) { set_error("ag_to_rgb_backward: skip gradient without taps"); return AG_ERR_INVALID_ARGUMENT; }
        if ((rc = skip_chain_backward_g(a->g_skip, a->g_out, a->skip_taps, G, a->Cout / 4, a->H / 2, a->W / 2, s))) return rc;
    }
    if (a->g_x && (rc = conv_backward_input_g(&d, G, a->g_out, a->Cout * hw, wm_t, a->g_x, a->Cin * hw, a->workspace, a->workspace_bytes, s))) return rc;
    if (a->g_weight) {
        if (!a->g_style) { set_error("ag_to_rgb_backward: g_weight without g_style"); return AG_ERR_INVALID_ARGUMENT; }
        float* g_wm = a->scratch;
        float* part = a->scratch + pad64((size_t)G * a->Cout * a->Cin);
        if ((rc = conv_backward_weight_g(&d, G, a->x, a->Cin * hw, a->g_out, a->Cout * hw, g_wm, (long long)a->Cout * a->Cin, a->workspace, a->workspace_bytes, s))) return rc;
        if ((rc = modulate_weight_backward_g(a->g_weight, a->g_style, part, g_wm, G, w_t, style_t, nullptr, a->scale, 0, a->Cout, a->Cin, 1, 0, s))) return rc;
    }
    return AG_OK;
}

// ---- comb convolution of a decoder stage without the concatenation (include/ag_layers.h AgGroupedCombArgs) ------------------------------
static bool comb_ok(const AgGroupedCombArgs* a)
{
    if (!a || a->M < 1 || a->M > AG_MAX_GROUPS || a->N < 1 || a->N > a->M || a->C1 <= 0 || a->C2 <= 0 || a->Cout <= 0 || a->H <= 0 || a->W <= 0) return false;
    if (a->member_begin[0] != 0 || a->member_begin[a->N] != a->M) return false;
    for (int r = 0; r < a->N; r++)
        if (a->member_begin[r + 1] <= a->member_begin[r]) return false;
    return ((long long)a->H * a->W * a->Cout) % 4 == 0;
}

static AgConvDesc comb_desc(const AgGroupedCombArgs* a, int cin)
{
    AgConvDesc d{};
    d.kind = AG_CONV; d.Cin = cin; d.Cout = a->Cout; d.H = a->H; d.W = a->W; d.k = 3; d.stride = 1; d.padding = 1; d.weight_scale = a->scale;
    return d;
}

size_t ag_grouped_comb_args_bytes(void) { return sizeof(AgGroupedCombArgs); }
size_t ag_grouped_comb_maxima_floats(void) { return conv_absmax_floats(4); }

size_t ag_grouped_comb_packed_bytes(const AgGroupedCombArgs* a, int32_t level_half)
{
    if (!comb_ok(a)) return 0;
    const AgConvDesc d = comb_desc(a, level_half ? a->C2 : a->C1);
    return conv_packed_bytes_g(&d, level_half ? a->N : a->M);
}

size_t ag_grouped_comb_workspace_bytes(const AgGroupedCombArgs* a)
{
    if (!comb_ok(a)) return 0;
    const AgConvDesc d = comb_desc(a, a->C1 > a->C2 ? a->C1 : a->C2);
    return conv_workspace_bytes_g(&d, a->M);
}

size_t ag_grouped_comb_scratch_floats(const AgGroupedCombArgs* a, int32_t backward)
{
    if (!comb_ok(a)) return 0;
    const size_t hw = (size_t)a->H * a->W;
    size_t n = pad64((size_t)a->M * a->Cout * hw) + pad64((size_t)a->N * a->Cout * hw);        // pre-activation (its gradient), the level half (its gradient)
    if (backward) n += pad64(noise_bias_act_partial_floats(a->M, a->Cout, (int)hw));
    n += pad64(conv_absmax_floats(6));            // fp16 split form: maxima of dy (members), dy (networks), x, lev, the two weight halves
    return n + 64;
}

int ag_grouped_comb_forward(const AgGroupedCombArgs* a, void* stream)
{
    if (!comb_ok(a) || !a->x || !a->lev || !a->out || !a->scratch) { set_error("ag_grouped_comb_forward: bad arguments"); return AG_ERR_INVALID_ARGUMENT; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long long hw = (long long)a->H * a->W;
    float* pre = a->scratch;
    float* t = a->scratch + pad64((size_t)a->M * a->Cout * hw);
    ConvOpts o;
    o.w_cin_total = a->C1 + a->C2;
    PtrTable w1{}, w2{}, addend{}, bias = table_of(a->act_bias, a->M);
    for (int r = 0; r < a->N; r++) {
        if (!a->weight[r]) { set_error("ag_grouped_comb_forward: null weight"); return AG_ERR_INVALID_ARGUMENT; }
        w2.p[r] = a->weight[r] + (size_t)a->C1 * 9;
        for (int m = a->member_begin[r]; m < a->member_begin[r + 1]; m++) { w1.p[m] = a->weight[r]; addend.p[m] = t + (size_t)r * a->Cout * hw; }
    }
    const AgConvDesc d1 = comb_desc(a, a->C1), d2 = comb_desc(a, a->C2);
    int rc;
    ConvOpts o1 = o, o2 = o;
    if (conv_math_needs_absmax()) {           // the operand maxima of both convolutions in one launch; kept for the backward when the caller has room
        float* am = a->operand_maxima ? a->operand_maxima : t + pad64((size_t)a->N * a->Cout * hw);
        const long long wrow = (long long)(a->C1 + a->C2) * 9;
        AmaxTensor mt[4] = { AmaxTensor{ a->x, nullptr, a->C1 * hw, a->C1 * hw, 0, 1, a->M }, AmaxTensor{ a->lev, nullptr, a->C2 * hw, a->C2 * hw, 0, 1, a->N },
                                   AmaxTensor{ nullptr, &w1, 0, (long long)a->C1 * 9, wrow, a->Cout, a->M }, AmaxTensor{ nullptr, &w2, 0, (long long)a->C2 * 9, wrow, a->Cout, a->N } };
        if (a->weights_cached && a->operand_maxima) mt[2] = mt[3] = AmaxTensor{};        // frozen weights: their slots are a previous call's
        if (a->x_maxima) mt[0] = AmaxTensor{ a->x_maxima, nullptr, kAmaxParts, kAmaxParts, 0, 1, a->M };   // handed over by the call that produced x: the slot
                                                                                                       // gets the handed maxima (as in ag_grouped_layer_forward)
        if ((rc = conv_absmax(mt, 4, a->M, am, s, a->out_maxima, a->M))) return rc;      // (and zeroes the slots of this call's own output maxima)
        const size_t slot = (size_t)kMaxGroups * kAmaxParts;
        o1.amax_x = a->x_maxima ? a->x_maxima : am; o2.amax_x = am + slot; o1.amax_w = am + 2 * slot; o2.amax_w = am + 3 * slot;
    }
    const bool frozen = a->weights_cached && (a->operand_maxima || !conv_math_needs_absmax());
    o1.packed = reinterpret_cast<float*>(a->packed_x); o1.packed_valid = frozen;
    o2.packed = reinterpret_cast<float*>(a->packed_lev); o2.packed_valid = frozen;
    float* const omax = (conv_math_needs_absmax() && a->out_maxima) ? a->out_maxima : nullptr;
    if ((rc = conv_forward_g(&d2, a->N, a->lev, a->C2 * hw, w2, nullptr, PtrTable{}, t, a->Cout * hw, a->workspace, a->workspace_bytes, s, o2))) return rc;
    if (fused_act()) {
        ConvAct act{ 2, a->slope, a->act_scale, addend, PtrTable{} };
        act.out_amax = omax;
        ConvOpts oa = o1;
        oa.act = &act;
        return conv_forward_g(&d1, a->M, a->x, a->C1 * hw, w1, nullptr, bias, a->out, a->Cout * hw, a->workspace, a->workspace_bytes, s, oa);
    }
    if ((rc = conv_forward_g(&d1, a->M, a->x, a->C1 * hw, w1, nullptr, PtrTable{}, pre, a->Cout * hw, a->workspace, a->workspace_bytes, s, o1))) return rc;
    if ((rc = bias_act_forward_addend_g(a->out, pre, a->M, addend, bias, a->Cout, (int)hw, a->slope, a->act_scale, s))) return rc;
    if (omax) {
        const AmaxTensor to{ a->out, nullptr, a->Cout * hw, a->Cout * hw, 0, 1 };
        return conv_absmax(&to, 1, a->M, omax, s);
    }
    return AG_OK;
}

int ag_grouped_comb_backward(const AgGroupedCombArgs* a, void* stream)
{
    if (!comb_ok(a) || !a->x || !a->lev || !a->out || !a->scratch || !a->g_out) { set_error("ag_grouped_comb_backward: bad arguments"); return AG_ERR_INVALID_ARGUMENT; }
    if (!a->g_weight && (a->g_weight_x == nullptr) != (a->g_weight_lev == nullptr)) { set_error("ag_grouped_comb_backward: both weight gradients or none"); return AG_ERR_INVALID_ARGUMENT; }
    const bool want_w = a->g_weight || a->g_weight_x;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long long hw = (long long)a->H * a->W;
    float* g_pre = a->scratch;
    float* g_t = a->scratch + pad64((size_t)a->M * a->Cout * hw);
    float* part = g_t + pad64((size_t)a->N * a->Cout * hw);
    ConvOpts o;
    o.w_cin_total = a->C1 + a->C2;
    PtrTable w1{}, w2{};
    for (int r = 0; r < a->N; r++) {
        if (!a->weight[r]) { set_error("ag_grouped_comb_backward: null weight"); return AG_ERR_INVALID_ARGUMENT; }
        w2.p[r] = a->weight[r] + (size_t)a->C1 * 9;
        for (int m = a->member_begin[r]; m < a->member_begin[r + 1]; m++) w1.p[m] = a->weight[r];
    }
    const AgConvDesc d1 = comb_desc(a, a->C1), d2 = comb_desc(a, a->C2);
    int rc;
    const bool maxima = conv_math_needs_absmax();
    float* am = part + pad64(noise_bias_act_partial_floats(a->M, a->Cout, (int)hw));
    if ((rc = noise_bias_act_backward_g(g_pre, a->g_out, a->out, a->M, PtrTable{}, a->g_bias, a->Cout, nullptr, 0, part, a->Cout, (int)hw, a->slope,
                                        a->act_scale, s, maxima ? am : nullptr))) return rc;            // (slot 0: max |g_pre| per member)
    const bool need_t = a->g_lev || want_w;
    if (need_t && (rc = sum_member_ranges(g_t, g_pre, a->member_begin, a->N, a->Cout * hw, s))) return rc;       // the level half sees the SUM of its members' gradients
    ConvOpts o1 = o, o2 = o, ow1, ow2;
    if (maxima) {       // the other operands of the four convolutions below, one launch
        const long long wrow = (long long)(a->C1 + a->C2) * 9;
        AmaxTensor mt[6] = { AmaxTensor{}, AmaxTensor{ g_t, nullptr, a->Cout * hw, a->Cout * hw, 0, 1, a->N },
                             AmaxTensor{ a->x, nullptr, a->C1 * hw, a->C1 * hw, 0, 1, a->M }, AmaxTensor{ a->lev, nullptr, a->C2 * hw, a->C2 * hw, 0, 1, a->N },
                             AmaxTensor{ nullptr, &w1, 0, (long long)a->C1 * 9, wrow, a->Cout, a->M }, AmaxTensor{ nullptr, &w2, 0, (long long)a->C2 * 9, wrow, a->Cout, a->N } };
        const bool kept = a->operand_maxima != nullptr;        // x, lev, w1, w2 in the forward's order
        if (!need_t) mt[1] = AmaxTensor{};
        if (!want_w || kept) mt[2] = mt[3] = AmaxTensor{};
        if (a->x_maxima) mt[2] = AmaxTensor{};
        if (!a->g_x || kept) mt[4] = AmaxTensor{};
        if (!a->g_lev || kept) mt[5] = AmaxTensor{};
        if ((rc = conv_absmax(mt, 6, a->M, am, s))) return rc;
        const size_t slot = (size_t)kMaxGroups * kAmaxParts;
        o1.amax_dy = ow1.amax_dy = am; o2.amax_dy = ow2.amax_dy = am + slot;
        ow1.amax_x = a->x_maxima ? a->x_maxima : kept ? a->operand_maxima : am + 2 * slot;
        ow2.amax_x = kept ? a->operand_maxima + slot : am + 3 * slot;
        o1.amax_w = kept ? a->operand_maxima + 2 * slot : am + 4 * slot;
        o2.amax_w = kept ? a->operand_maxima + 3 * slot : am + 5 * slot;
    }
    if (a->g_x && (rc = conv_backward_input_g(&d1, a->M, g_pre, a->Cout * hw, w1, a->g_x, a->C1 * hw, a->workspace, a->workspace_bytes, s, o1))) return rc;
    if (a->g_lev && (rc = conv_backward_input_g(&d2, a->N, g_t, a->Cout * hw, w2, a->g_lev, a->C2 * hw, a->workspace, a->workspace_bytes, s, o2))) return rc;
    if (a->g_weight) {
        // both halves straight into the parameters' gradients [N][Cout][C1 + C2][3][3]: the members of a network are summed into its tensor's first
        // C1 channels (consecutive instances with one destination: the weight gradient's fixed-order slice reduction adds them), the level half
        // fills the other C2; every element is written, nothing is zeroed first
        const long long wrow = (long long)(a->C1 + a->C2) * 9, wnet = (long long)a->Cout * wrow;
        PtrTable t1{}, t2{};
        for (int r = 0; r < a->N; r++) {
            t2.p[r] = a->g_weight + (size_t)r * wnet + (size_t)a->C1 * 9;
            for (int m = a->member_begin[r]; m < a->member_begin[r + 1]; m++) t1.p[m] = a->g_weight + (size_t)r * wnet;
        }
        ow1.dw_table = &t1; ow2.dw_table = &t2;
        ow1.dw_row_stride = ow2.dw_row_stride = wrow;
        if ((rc = conv_backward_weight_g(&d1, a->M, a->x, a->C1 * hw, g_pre, a->Cout * hw, nullptr, 0, a->workspace, a->workspace_bytes, s, ow1))) return rc;
        if ((rc = conv_backward_weight_g(&d2, a->N, a->lev, a->C2 * hw, g_t, a->Cout * hw, nullptr, 0, a->workspace, a->workspace_bytes, s, ow2))) return rc;
    } else if (a->g_weight_x) {
        if ((rc = conv_backward_weight_g(&d1, a->M, a->x, a->C1 * hw, g_pre, a->Cout * hw, a->g_weight_x, (long long)a->Cout * a->C1 * 9, a->workspace,
                                         a->workspace_bytes, s, ow1))) return rc;
        if ((rc = conv_backward_weight_g(&d2, a->N, a->lev, a->C2 * hw, g_t, a->Cout * hw, a->g_weight_lev, (long long)a->Cout * a->C2 * 9, a->workspace,
                                         a->workspace_bytes, s, ow2))) return rc;
    }
    return AG_OK;
}

/* Stacked copies of the per-instance helpers the grouped network needs between layers. */
int ag_grouped_block2x2(float* out, const float* in, const float* matrix16, int32_t merge, int32_t G, int32_t C, int32_t h, int32_t w, void* stream)
{
    return block2x2_transform_g(out, in, matrix16, merge, G, C, h, w, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
