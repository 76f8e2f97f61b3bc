// Layer-level entry points (include/ag_layers.h): one native call per ConvLayer / StyledConv, forward and backward.  Pure composition of
// the per-kernel entry points -- same kernels, same order, same results -- so that the host pays one Python -> C transition per layer
// instead of three to five.
#include "ag_common.h"
#include "../../include/ag_conv.h"
#include "../../include/ag_layers.h"
#include "../../include/ag_styleunet.h"

#include <cstdlib>

namespace {

// Blur + noise / bias / activation of an up-sampling StyledConv as ONE pass each way (ag_fir4x4_noise_bias_act_*): AG_FUSED_TAIL bit 0
// forward, bit 1 backward.  OFF by default: built, bit-identical to the two passes (tests/test_styleunet_ops.py, test_styleunet_net.py),
// and measured same-box with profiles/ab_tail.sh -- network forward + backward 17.5-17.6 ms with two passes, 17.7-17.8 with the fused
// forward, 18.2-18.8 with both fused.  It removes 12 of the ~1500 launches of a network pass; the fused backward (activation gradient
// formed on the fly inside the FIR's adjoint, 70 loads per thread) is slower than the streaming pass + FIR it replaces.
int fused_tail_mask()
{
    static const int m = [] { const char* e = getenv("AG_FUSED_TAIL"); return e ? atoi(e) : 0; }();
    return m;
}

struct Geo {
    int OH, OW;          // layer output
    int CH, CW;          // convolution output (before the Blur of an up-sampling StyledConv)
    int BH, BW;          // blurred input of a down-sampling ConvLayer
    AgConvDesc d;
};

bool geometry(const AgLayerArgs* a, Geo& g)
{
    if (!a || a->Cin <= 0 || a->Cout <= 0 || a->H <= 0 || a->W <= 0 || a->k <= 0) return false;
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

}  // namespace

extern "C" {

size_t ag_layer_args_bytes(void) { return sizeof(AgLayerArgs); }

int ag_layer_output_size(const AgLayerArgs* a, int32_t* OH, int32_t* OW)
{
    Geo g;
    if (!geometry(a, g) || !OH || !OW) { ag::set_error("ag_layer_output_size: bad layer description"); return AG_ERR_INVALID_ARGUMENT; }
    *OH = g.OH; *OW = g.OW;
    return AG_OK;
}

size_t ag_layer_scratch_floats(const AgLayerArgs* a, int32_t backward)
{
    Geo g;
    if (!geometry(a, g)) return 0;
    const size_t pre = (size_t)a->Cout * g.OH * g.OW;                         // activation input / its gradient
    size_t n = pre + 64;
    if (a->modulated && a->resample) n += (size_t)a->Cout * g.CH * g.CW + 64;   // transposed-convolution output / its gradient
    if (!a->modulated && a->resample && backward) n += (size_t)a->Cin * g.BH * g.BW + 64;   // gradient of the blurred input
    if (backward && a->modulated) n += (size_t)a->Cout * a->Cin * a->k * a->k + 64;   // gradient of the modulated weight
    return n;
}

int ag_layer_forward(const AgLayerArgs* a, void* stream)
{
    Geo g;
    if (!geometry(a, g) || !a->x || !a->weight || !a->out || !a->scratch) { ag::set_error("ag_layer_forward: bad arguments"); return AG_ERR_INVALID_ARGUMENT; }
    const size_t pre_n = ((size_t)a->Cout * g.OH * g.OW + 63) / 64 * 64;
    float* pre = a->scratch;
    float* aux = a->scratch + pre_n;
    int rc;
    if (!a->modulated) {
        const float* cx = a->x;
        if (a->resample) {
            if (!a->k_blur || !a->x_blur) { ag::set_error("ag_layer_forward: down-sampling layer without FIR taps / x_blur"); return AG_ERR_INVALID_ARGUMENT; }
            if ((rc = ag_upfirdn2d(a->x_blur, a->x, a->k_blur, a->Cin, a->H, a->W, 4, 4, 1, 1, 1, 1, 2, 2, 2, 2, stream))) return rc;
            cx = a->x_blur;
        }
        if ((rc = ag_conv_forward(&g.d, cx, a->weight, nullptr, nullptr, pre, a->workspace, a->workspace_bytes, stream))) return rc;
        return ag_noise_bias_act_forward(a->out, pre, nullptr, nullptr, a->act_bias, a->Cout, g.OH * g.OW, a->slope, a->act_scale, stream);
    }
    if (!a->style || !a->w_mod || !a->demod) { ag::set_error("ag_layer_forward: StyledConv needs style, w_mod and demod"); return AG_ERR_INVALID_ARGUMENT; }
    if ((rc = ag_modulate_weight_forward(a->w_mod, a->demod, a->weight, a->style, a->scale, 1, a->Cout, a->Cin, a->k * a->k, a->resample ? 1 : 0, stream))) return rc;
    if (a->resample) {
        if (!a->k_blur) { ag::set_error("ag_layer_forward: resampling layer without FIR taps"); return AG_ERR_INVALID_ARGUMENT; }
        if ((rc = ag_conv_forward(&g.d, a->x, a->w_mod, nullptr, nullptr, aux, a->workspace, a->workspace_bytes, stream))) return rc;
        // Blur + noise + bias + activation in one pass (round 3); the filtered, pre-activation tensor never goes to memory
        const bool nzr = a->noise && a->noise_weight;
        if (!(fused_tail_mask() & 1)) {
            if ((rc = ag_upfirdn2d(pre, aux, a->k_blur, a->Cout, g.CH, g.CW, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1, stream))) return rc;
            return ag_noise_bias_act_forward(a->out, pre, nzr ? a->noise : nullptr, nzr ? a->noise_weight : nullptr, a->act_bias, a->Cout,
                                             g.OH * g.OW, a->slope, a->act_scale, stream);
        }
        return ag_fir4x4_noise_bias_act_forward(a->out, aux, a->k_blur, a->Cout, g.CH, g.CW, 1, 1, nzr ? a->noise : nullptr,
                                                nzr ? a->noise_weight : nullptr, a->act_bias, a->slope, a->act_scale, stream);
    } else {
        if ((rc = ag_conv_forward(&g.d, a->x, a->w_mod, nullptr, nullptr, pre, a->workspace, a->workspace_bytes, stream))) return rc;
    }
    const bool nz = a->noise && a->noise_weight;
    return ag_noise_bias_act_forward(a->out, pre, nz ? a->noise : nullptr, nz ? a->noise_weight : nullptr, a->act_bias, a->Cout, g.OH * g.OW,
                                     a->slope, a->act_scale, stream);
}

int ag_layer_backward(const AgLayerArgs* a, void* stream)
{
    Geo g;
    if (!geometry(a, g) || !a->x || !a->weight || !a->out || !a->scratch || !a->g_out) { ag::set_error("ag_layer_backward: bad arguments"); return AG_ERR_INVALID_ARGUMENT; }
    const size_t pre_n = ((size_t)a->Cout * g.OH * g.OW + 63) / 64 * 64;
    float* g_pre = a->scratch;
    float* aux = a->scratch + pre_n;
    const bool nz = a->modulated && a->noise && a->noise_weight;
    float* gb = (a->want_bias && a->g_bias_noise) ? a->g_bias_noise : nullptr;
    float* gnw = (nz && a->want_noise_weight && a->g_bias_noise) ? a->g_bias_noise + a->Cout : nullptr;
    int rc;
    const bool fused_tail = a->modulated && a->resample && (fused_tail_mask() & 2);       // activation backward + the Blur's adjoint as one pass (below)
    if (!fused_tail &&
        (rc = ag_noise_bias_act_backward(g_pre, a->g_out, a->out, gnw ? a->noise : nullptr, gb, gnw, a->Cout, g.OH * g.OW, a->slope, a->act_scale, stream))) return rc;
    if (!a->modulated) {
        const float* cx = a->x;
        if (a->resample) {
            if (!a->x_blur || !a->k_blur) { ag::set_error("ag_layer_backward: down-sampling layer without x_blur / FIR taps"); return AG_ERR_INVALID_ARGUMENT; }
            if (a->g_x) {     // gradient w.r.t. the blurred input, then the FIR's adjoint (flipped taps, pads (1,1): [H + 1] -> [H])
                if ((rc = ag_conv_backward_input(&g.d, g_pre, a->weight, aux, a->workspace, a->workspace_bytes, stream))) return rc;
                if ((rc = ag_upfirdn2d(a->g_x, aux, a->k_blur, a->Cin, g.BH, g.BW, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1, stream))) return rc;
            }
            if (a->g_weight && (rc = ag_conv_backward_weight(&g.d, a->x_blur, g_pre, a->g_weight, a->workspace, a->workspace_bytes, stream))) return rc;
            return AG_OK;
        }
        if (a->g_x && (rc = ag_conv_backward_input(&g.d, g_pre, a->weight, a->g_x, a->workspace, a->workspace_bytes, stream))) return rc;
        if (a->g_weight && (rc = ag_conv_backward_weight(&g.d, cx, g_pre, a->g_weight, a->workspace, a->workspace_bytes, stream))) return rc;
        return AG_OK;
    }
    if (!a->style || !a->w_mod || !a->demod) { ag::set_error("ag_layer_backward: StyledConv needs style, w_mod and demod"); return AG_ERR_INVALID_ARGUMENT; }
    const float* g_conv = g_pre;
    float* after = aux;
    if (a->resample) {
        // activation backward + adjoint of Blur pad (1,1) (pads (2,2) with the flipped taps: [2H] -> [2H + 1]) in one pass
        if (!fused_tail) {
            if ((rc = ag_upfirdn2d(aux, g_pre, a->k_blur, a->Cout, g.OH, g.OW, 4, 4, 1, 1, 1, 1, 2, 2, 2, 2, stream))) return rc;
        } else if ((rc = ag_fir4x4_noise_bias_act_backward(aux, a->g_out, a->out, a->k_blur, a->Cout, g.OH, g.OW, gnw ? a->noise : nullptr, gb, gnw,
                                                    a->slope, a->act_scale, stream))) return rc;
        g_conv = aux;
        after = aux + ((size_t)a->Cout * g.CH * g.CW + 63) / 64 * 64;
    }
    if (a->g_x && (rc = ag_conv_backward_input(&g.d, g_conv, a->w_mod, a->g_x, a->workspace, a->workspace_bytes, stream))) return rc;
    if (a->g_weight) {
        if (!a->g_style) { ag::set_error("ag_layer_backward: g_weight without g_style"); return AG_ERR_INVALID_ARGUMENT; }
        float* g_wm = after;
        if ((rc = ag_conv_backward_weight(&g.d, a->x, g_conv, g_wm, a->workspace, a->workspace_bytes, stream))) return rc;
        if ((rc = ag_modulate_weight_backward(a->g_weight, a->g_style, g_wm, a->weight, a->style, a->demod, a->scale, 1, a->Cout, a->Cin, a->k * a->k,
                                              a->resample ? 1 : 0, stream))) return rc;
    }
    return AG_OK;
}

}  // extern "C"
