/*
 * ag_optim.h -- the optimizer step of a training iteration (round 5).
 *
 * The reference trains AvatarNet with torch.optim.Adam over all 224 M parameters of its three StyleUNets (main_avatar.py:60-66 builds it,
 * :248-250 steps it once per iteration).  torch's own fused multi-tensor kernel moves the step's 28 bytes per parameter at 3.6 TB/s on MI355X
 * (1.74 ms of a 34-ms iteration); this one streams 16 bytes per lane and load at ~5.5 TB/s.  Same arithmetic, same state (exp_avg, exp_avg_sq,
 * step count): torch.optim.Adam's update with amsgrad off,
 *     g      = grad (+ weight_decay * param) (maximize: -grad)
 *     m      = m + (1 - beta1) * (g - m)
 *     v      = beta2 * v + (1 - beta2) * g * g
 *     param -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 * in fp32; the two bias corrections are computed by the caller in double, per tensor (a parameter that received no gradient in some iterations --
 * the reference's pretraining pass leaves the colour network out -- has taken fewer steps), and passed in.
 */
#ifndef AG_OPTIM_H
#define AG_OPTIM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AG_ADAM_MAX_TENSORS 48

typedef struct AgAdamArgs {
    int32_t n;                                    /* tensors in this call, 1 .. AG_ADAM_MAX_TENSORS */
    int32_t maximize;
    float* param[AG_ADAM_MAX_TENSORS];            /* updated in place */
    const float* grad[AG_ADAM_MAX_TENSORS];
    float* exp_avg[AG_ADAM_MAX_TENSORS];          /* updated in place */
    float* exp_avg_sq[AG_ADAM_MAX_TENSORS];       /* updated in place */
    int64_t numel[AG_ADAM_MAX_TENSORS];
    float lr, beta1, beta2, eps, weight_decay;
    float one_minus_beta1, one_minus_beta2;       /* formed by the caller in double: 1 - 0.999f in fp32 is off by 1.3e-5 of its value */
    float bias_correction1[AG_ADAM_MAX_TENSORS];        /* per tensor: 1 - beta1^t (t = the steps THAT tensor has taken, this one included:   */
    float bias_correction2_sqrt[AG_ADAM_MAX_TENSORS];   /* sqrt(1 - beta2^t)        torch.optim.Adam counts steps per parameter)              */
} AgAdamArgs;

size_t ag_adam_args_bytes(void);
/* One launch for all `n` tensors (dense fp32, any alignment / length).  Returns AG_OK or an AG_ERR_* code (include/ag_raster.h). */
int ag_adam_step(const AgAdamArgs* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AG_OPTIM_H */
